// roll_inst.hip — explicit instantiations of the rolling-accumulator kernels (dog_roll.hpp) for the kernel
// lengths of one PDOG_ROLL_SET (roll_lengths.def).  Compiled once per set so that the sets build in parallel;
// pawsome_dog.hip declares the same instantiations `extern` and only takes their addresses.
#define PDOG_ROLL_INST_ONLY // no private copies of the static kernels (finish, mode, chain step): pawsome_dog.hip holds them
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
// l = 65 (target_width 25, the reference default) for EVERY window-height class (roll_epi_class = ((n1 + 2) ÷ 4) mod 18):
// statically shortened epilogue bodies, two classes per set.  window_size → rows → class, the common ones first:
// 256 → 257 → 10, 512 → 513 → 2, 64 → 65 → 16, 128 → 129 → 14, 384 → 385 → 6, 1024 → 1025 → 4.
#define PDOG_EPI_INST(C) template __global__ void dog_roll_kernel<65, false, 0, C>(const LaunchGeo, const f2 *, const f2 *);
#if PDOG_ROLL_SET == 9
PDOG_EPI_INST(10) PDOG_EPI_INST(2)
#endif
#if PDOG_ROLL_SET == 10
PDOG_EPI_INST(16) PDOG_EPI_INST(14)
#endif
#if PDOG_ROLL_SET == 11
PDOG_EPI_INST(6) PDOG_EPI_INST(4)
#endif
#if PDOG_ROLL_SET == 12
PDOG_EPI_INST(0) PDOG_EPI_INST(1)
#endif
#if PDOG_ROLL_SET == 13
PDOG_EPI_INST(3) PDOG_EPI_INST(5)
#endif
#if PDOG_ROLL_SET == 14
PDOG_EPI_INST(7) PDOG_EPI_INST(8)
#endif
#if PDOG_ROLL_SET == 15
PDOG_EPI_INST(9) PDOG_EPI_INST(11)
#endif
#if PDOG_ROLL_SET == 16
PDOG_EPI_INST(12) PDOG_EPI_INST(13)
#endif
#if PDOG_ROLL_SET == 17
PDOG_EPI_INST(15) PDOG_EPI_INST(17)
#endif
#undef PDOG_EPI_INST
} // namespace pdog
