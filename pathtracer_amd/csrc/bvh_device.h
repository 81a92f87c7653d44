// internal: GPU BVH build (bvh_device.hip), wrapped by pt_build_bvh_device in pt355.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

// 0 = ok, 1 = bad argument, 2 = HIP error; `err` receives the text
// method 0 = LBVH (radix tree over Morton keys), 1 = PLOC (bottom-up clustering by merged-box area; falls back to 0 when its tree
// comes out deeper than the traversal stack)
int ptb_build_bvh(hipStream_t stream, int method, const void *tris_host, uint32_t n, uint32_t max_leaf, void *nodes_out, uint32_t *n_nodes_out,
                  void *sorted_out, float *build_ms, char *err, size_t errlen);
