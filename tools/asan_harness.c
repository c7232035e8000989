#include "pawsome_dog.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void) {
    srand(7);
    for (int it = 0; it < 3000; ++it) {
        int fh = 1 + rand() % 90, fw = 1 + rand() % 120, pad = rand() % 9;
        double tws[] = {2, 5, 10, 16, 25, 40};
        double tw = tws[rand() % 6];
        int wh = 1 + rand() % 70, ww = 1 + rand() % 70;
        int l = pdog_kernel_len(tw), hw = l / 2, r1 = wh / 2, r2 = ww / 2;
        int th = 2 * r1 + l, tww = 2 * r2 + l;
        int64_t stride = fw + pad, pitch = tww + rand() % 20;
        uint8_t *frame = malloc((size_t)fh * stride - pad); /* exactly the bytes a strided frame owns */
        for (size_t i = 0; i < (size_t)fh * stride - pad; ++i) frame[i] = rand();
        uint8_t *out = malloc((size_t)th * pitch);
        int32_t g[2] = {-hw + rand() % (fh + 2 * hw + 2), -hw + rand() % (fw + 2 * hw + 2)};
        if (pdog_window_tile(frame, fh, fw, stride, rand() % 256, tw, wh, ww, g, out, pitch) != PDOG_OK) { printf("fail %s\n", pdog_last_error()); return 1; }
        int m;
        if (pdog_mode_u8(frame, fh, fw, stride, &m) != PDOG_OK) return 2;
        free(frame); free(out);
    }
    double taps[400];
    if (pdog_gaussian_taps(120.0, 1, taps, 400) != PDOG_OK) return 3;
    printf("asan harness ok\n");
    return 0;
}
