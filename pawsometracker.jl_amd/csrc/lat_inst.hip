// lat_inst.hip — explicit instantiations of the latency kernels' compile-time-length instances (dog_fused.hpp,
// dog_tiled.hpp) for ONE kernel length, -DPDOG_LAT_L=<l> (lat_lengths.def): a translation unit per length so that they
// build in parallel; pawsome_dog.hip declares the same instantiations `extern` and only takes their addresses.
#define PDOG_ROLL_INST_ONLY // no private copies of the static kernels (finish, mode, chain step, publish): pawsome_dog.hip holds them
#include "dog_tiled.hpp"
#ifndef PDOG_LAT_L
#error "compile with -DPDOG_LAT_L=<kernel length>"
#endif
namespace pdog {
template __global__ void dog_fused_kernel<false, 0, PDOG_LAT_L>(const FusedGeo, const f2 *, const f2 *);
template __global__ void dog_fused_kernel<true, 0, PDOG_LAT_L>(const FusedGeo, const f2 *, const f2 *);
template __global__ void dog_tiled_kernel<false, PDOG_LAT_L>(const TiledGeo, const f2 *, const f2 *);
template __global__ void dog_tiled_kernel<true, PDOG_LAT_L>(const TiledGeo, const f2 *, const f2 *);
} // namespace pdog
