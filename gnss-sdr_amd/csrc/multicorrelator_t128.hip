// The batched multicorrelator once more, with work-groups of 128 threads: TWO waves per job (csrc/multicorrelator.hip, whose source this is, explains when
// mcorr_launch takes these kernels).  mcorr_device.h keeps the two sizes in separate namespaces (mcdev_256 / mcdev_128); the kernels are named
// mcorr_kernel_t128<...> so that a profile tells them apart.
#define GSH_MC_THREADS 128
#define GSH_MC_VARIANT_128 1
#include "multicorrelator.hip"
