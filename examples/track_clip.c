/* examples/track_clip.c — the reference's frame loop (src/PawsomeTracker.jl:159-169) in plain C over the C ABI:
 *   Tracker(img, target_width, window_size, darker_target)  ->  pdog_mode_u8 + pdog_create     (:39-52)
 *   read!(vid, trckr.img.data); ij = trckr(ij)              ->  pdog_detect_host per frame      (:166-167)
 * on a synthetic clip (a dark disc on a mid-grey background that walks across the frame, the recipe of
 * test/test-basic-test.jl:65-68).  Prints one "row col" line per frame; exit code 0 when every position is
 * within one pixel of the disc centre.
 *   gcc -std=c99 -I include examples/track_clip.c -L pawsometracker.jl_amd -l:libpawsome_dog.so -lm -o track_clip */
#include "pawsome_dog.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void draw(uint8_t *f, int h, int w, int ci, int cj, int rad)
{
    memset(f, 128, (size_t)h * w);
    for (int i = ci - rad; i <= ci + rad; ++i)
        for (int j = cj - rad; j <= cj + rad; ++j)
            if (i >= 1 && i <= h && j >= 1 && j <= w && (i - ci) * (i - ci) + (j - cj) * (j - cj) <= rad * rad)
                f[(size_t)(i - 1) * w + (j - 1)] = 0;
}

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != PDOG_OK) {                                              \
            fprintf(stderr, "%s: %s\n", #call, pdog_last_error());         \
            return 2;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    const int h = 240, w = 320, n_frames = argc > 1 ? atoi(argv[1]) : 50;
    const double target_width = 25.0;
    const int win = pdog_default_window(target_width); /* guess_window_size, :64-68 */
    uint8_t *frame = (uint8_t *)malloc((size_t)h * w);
    int ci = 60, cj = 40, fill = 0, bad = 0;
    draw(frame, h, w, ci, cj, (int)target_width / 2);
    CHECK(pdog_mode_u8(frame, h, w, w, &fill)); /* mode(_img), :47 */
    pdog_tracker *t = NULL;
    CHECK(pdog_create(0, h, w, target_width, win, win, 1, fill, &t));
    int32_t ij[2] = {ci + 4, cj - 3}; /* start_location */
    for (int k = 0; k < n_frames; ++k) {
        ci += 3; cj += 5;          /* the target moves less than the window radius per frame */
        if (ci > h - 20) ci = 20;  /* … except here: a jump the tracker cannot follow is not part of the demo */
        if (cj > w - 20) { cj = 20; ij[1] = cj; }
        if (ci == 20) ij[0] = ci;
        draw(frame, h, w, ci, cj, (int)target_width / 2);
        int32_t out[2];
        CHECK(pdog_detect_host(t, frame, w, ij, out, NULL));
        printf("%d %d\n", out[0], out[1]);
        if (abs(out[0] - ci) > 1 || abs(out[1] - cj) > 1) ++bad;
        ij[0] = out[0]; ij[1] = out[1];
    }
    CHECK(pdog_destroy(t));
    free(frame);
    return bad ? 1 : 0;
}
