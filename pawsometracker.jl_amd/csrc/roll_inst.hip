// roll_inst.hip — explicit instantiations of the rolling-accumulator kernels (dog_roll.hpp) for the kernel
// lengths of one PDOG_ROLL_SET (roll_lengths.def).  Compiled once per set so that the sets build in parallel;
// pawsome_dog.hip declares the same instantiations `extern` and only takes their addresses.
#include "dog_roll.hpp"
#ifndef PDOG_ROLL_SET
#error "compile with -DPDOG_ROLL_SET=<n>"
#endif
namespace pdog {
#define PDOG_ROLL_L(LT)                                                                                   \
    template __global__ void dog_roll_kernel<LT, false, 0>(const LaunchGeo, const f2 *, const f2 *);      \
    template __global__ void dog_roll_kernel<LT, true, 0>(const LaunchGeo, const f2 *, const f2 *);       \
    template __global__ void dog_thin_kernel<LT, false>(const LaunchGeo, const f2 *, const f2 *);         \
    template __global__ void dog_thin_kernel<LT, true>(const LaunchGeo, const f2 *, const f2 *);          \
    template __global__ void dog_chain_kernel<LT>(const ChainGeo, const f2 *, const f2 *);
#include "roll_lengths.def"
#undef PDOG_ROLL_L
#if PDOG_ROLL_SET == 9
// l = 65 (target_width 25, the reference default) for the window-height classes of the common window sizes
// (roll_epi_class: 256 → 257 rows = class 10, 512 → 513 rows = class 2): statically shortened epilogue bodies
template __global__ void dog_roll_kernel<65, false, 0, 10>(const LaunchGeo, const f2 *, const f2 *);
template __global__ void dog_roll_kernel<65, false, 0, 2>(const LaunchGeo, const f2 *, const f2 *);
#endif
} // namespace pdog
