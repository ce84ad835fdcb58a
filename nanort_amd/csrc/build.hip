// placeholder — replaced by the GPU binned-SAH builder
#include <string>
#include "common.h"
namespace nrt {
struct BuildResult {
  uint64_t num_nodes;
  uint32_t max_depth, num_leaves, num_branches;
};
template <typename T>
hipError_t gpu_build(int, hipStream_t, const T *, const uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t,
                     typename Wire<T>::Node **, uint32_t **, BuildResult *, std::string *err) {
  *err = "GPU build not implemented yet";
  return hipSuccess;
}
template hipError_t gpu_build<float>(int, hipStream_t, const float *, const uint32_t *, uint32_t, uint32_t, uint32_t,
                                     uint32_t, nrt_node_f32 **, uint32_t **, BuildResult *, std::string *);
template hipError_t gpu_build<double>(int, hipStream_t, const double *, const uint32_t *, uint32_t, uint32_t, uint32_t,
                                      uint32_t, nrt_node_f64 **, uint32_t **, BuildResult *, std::string *);
}
