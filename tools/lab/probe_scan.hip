// tools/lab/probe_scan.hip -- compiles ONE instantiation of the scan kernel and prints its register / spill figures:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I postgres-word2vec_amd/csrc -c -o /dev/null \
//         tools/lab/probe_scan.hip -Rpass-analysis=kernel-resource-usage
#include "fused5.h"
using namespace freddy;
const void* probe_kernels[] = {(const void*)&ivf_filter5_kernel<12, true, false>};
