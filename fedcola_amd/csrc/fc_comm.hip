// Server-side aggregation behind the C ABI: RCCL communicator, closed-form blend + all-reduce, and the exact-order
// (all-gather + sequential blend) verification mode.  Reference: FedavgServer._aggregate, src/server/fedavgserver.py:591-668
// (the in-place loop `g += (theta_i - g) * c_i` over the sampled clients, :656-664) and the per-round call site :812-819.
//
// RCCL is bound at run time (dlopen of librccl.so, the copy the process already maps -- PyTorch's -- if there is one), so the
// library loads on a machine without RCCL and a caller that never aggregates across processes never touches it.  A process
// is one rank = one GPU; xGMI is RCCL's business.
#include <dlfcn.h>

#include <mutex>
#include <stdlib.h>
#include <string.h>

#include "../../include/fedcola_hip.h"
#include "fc_kernels.h"

// ---------------------------------------------------------------- RCCL binding (rccl.h, ROCm 7.2)
typedef struct { char internal[128]; } fc_nccl_uid;      // ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128
typedef void* fc_nccl_comm;
enum { FC_NCCL_FLOAT32 = 7, FC_NCCL_SUM = 0 };
struct RcclApi {
  void* so = nullptr;
  int (*GetUniqueId)(fc_nccl_uid*) = nullptr;
  int (*CommInitRank)(fc_nccl_comm*, int, fc_nccl_uid, int) = nullptr;
  int (*CommDestroy)(fc_nccl_comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, fc_nccl_comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, fc_nccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static void rccl_load(RcclApi& api);
static RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;                       // two threads may create communicators at the same time
  std::call_once(once, [] { rccl_load(api); });
  return api.so ? &api : nullptr;
}
static void rccl_load(RcclApi& api) {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  for (const char* n : names) {                     // already mapped (e.g. by torch)?  use that copy
    so = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (so) break;
  }
  for (size_t i = 0; !so && i < sizeof(names) / sizeof(names[0]); ++i) so = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!so) return;
#define FC_SYM(field, name) *(void**)(&api.field) = dlsym(so, name)
  FC_SYM(GetUniqueId, "ncclGetUniqueId");
  FC_SYM(CommInitRank, "ncclCommInitRank");
  FC_SYM(CommDestroy, "ncclCommDestroy");
  FC_SYM(AllReduce, "ncclAllReduce");
  FC_SYM(AllGather, "ncclAllGather");
  FC_SYM(GetErrorString, "ncclGetErrorString");
#undef FC_SYM
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather) return;
  api.so = so;
}
#define FC_CHECK_NCCL(expr)                                                                                  \
  do {                                                                                                       \
    int _r = (expr);                                                                                         \
    if (_r != 0) {                                                                                           \
      fc_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "RCCL error"); \
      return -3;                                                                                             \
    }                                                                                                        \
  } while (0)

struct fc_comm {
  fc_nccl_comm comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" int fc_comm_unique_id(void* id_out, size_t bytes) {
  FC_REQUIRE(id_out && bytes >= FC_COMM_ID_BYTES, "fc_comm_unique_id: need a %d-byte buffer", FC_COMM_ID_BYTES);
  FC_REQUIRE(rccl(), "fc_comm_unique_id: librccl.so could not be loaded");
  fc_nccl_uid id;
  FC_CHECK_NCCL(rccl()->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}
extern "C" int fc_comm_create(const void* id, size_t bytes, int32_t rank, int32_t world, fc_comm_t** out) {
  FC_REQUIRE(id && out && bytes >= FC_COMM_ID_BYTES, "fc_comm_create: null argument / short id");
  FC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "fc_comm_create: bad rank %d of %d", rank, world);
  FC_REQUIRE(rccl(), "fc_comm_create: librccl.so could not be loaded");
  fc_nccl_uid uid;
  memcpy(&uid, id, sizeof(uid));
  fc_comm* c = new fc_comm();
  c->rank = rank; c->world = world;
  int r = rccl()->CommInitRank(&c->comm, world, uid, rank);
  if (r != 0) {
    fc_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error");
    delete c;
    return -3;
  }
  *out = c;
  return 0;
}
extern "C" void fc_comm_destroy(fc_comm_t* c) {
  if (!c) return;
  if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
  delete c;
}
extern "C" int32_t fc_comm_rank(const fc_comm_t* c) { return c ? c->rank : 0; }
extern "C" int32_t fc_comm_world(const fc_comm_t* c) { return c ? c->world : 1; }

extern "C" int fc_allreduce_sum(fc_comm_t* c, float* buf, int64_t n, void* stream) {
  if (!c || n <= 0) return 0;
  FC_REQUIRE(buf, "fc_allreduce_sum: null buffer");
  FC_CHECK_NCCL(rccl()->AllReduce(buf, buf, (size_t)n, FC_NCCL_FLOAT32, FC_NCCL_SUM, c->comm, (hipStream_t)stream));
  return 0;
}

// ---------------------------------------------------------------- closed-form blend with the client pointers as kernel arguments
#define FC_AGG_MAX_CLIENTS 64
struct AggBases { const float* p[FC_AGG_MAX_CLIENTS]; };
// Both blend kernels: grid (AGG_BLOCKS_X, segments); a block strides its segment in 16-byte vectors (every stream of a segment -- global,
// clients at their own offsets, output -- is read / written as float4 when all of them are 16-byte aligned, which the 64-element segment
// alignment of fc_model_segment guarantees for models of one family), a scalar loop takes the tail or an unaligned segment.  HBM-bound:
// (participating clients + 2) x 4 bytes per element.
#define AGG_BLOCKS_X 64
__device__ __forceinline__ bool agg_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
__global__ void __launch_bounds__(256) k_blend_v(float* out, const float* g, AggBases bases, int m,
                                                 const int64_t* __restrict__ seg_off, const int64_t* __restrict__ seg_len,
                                                 const int64_t* __restrict__ src_off, const float* __restrict__ seg_w) {
  const int sgi = blockIdx.y;
  const int64_t off = seg_off[sgi], len = seg_len[sgi];
  const float* w = seg_w + (size_t)sgi * (m + 1);
  const int64_t* so = src_off + (size_t)sgi * m;
  const float wg = w[0];
  bool vec = agg_al16(out + off) && agg_al16(g + off);
  for (int j = 0; j < m; ++j)
    if (w[1 + j] != 0.f && so[j] >= 0) vec = vec && agg_al16(bases.p[j] + so[j]);
  const int64_t n4 = vec ? len >> 2 : 0;
  const float4* g4 = (const float4*)(g + off);
  float4* o4 = (float4*)(out + off);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wg != 0.f) { const float4 x = g4[i]; acc = make_float4(wg * x.x, wg * x.y, wg * x.z, wg * x.w); }
    for (int j = 0; j < m; ++j) {
      const float wj = w[1 + j];
      if (wj != 0.f && so[j] >= 0) {
        const float4 x = ((const float4*)(bases.p[j] + so[j]))[i];
        acc.x += wj * x.x; acc.y += wj * x.y; acc.z += wj * x.z; acc.w += wj * x.w;
      }
    }
    o4[i] = acc;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
    float acc = wg != 0.f ? wg * g[off + i] : 0.f;
    for (int j = 0; j < m; ++j) {
      const float wj = w[1 + j];
      if (wj != 0.f && so[j] >= 0) acc += wj * bases.p[j][so[j] + i];
    }
    out[off + i] = acc;
  }
}
// sequential blend in the reference's order and rounding: g <- g + fl32((theta_j - g) * c_j), j ascending (fedavgserver.py:656-664);
// every product and sum rounded on its own (no multiply-add contraction), component by component in the vector form: the same bits
__device__ __forceinline__ float blend_seq1(float acc, float x, float cj) {
#pragma clang fp contract(off)      // hipcc contracts a*b+c into an fma by default (also through __fmul_rn / __fadd_rn): three roundings here
  const float d = x - acc;
  const float p = d * cj;
  return acc + p;
}
__global__ void __launch_bounds__(256) k_blend_seq(float* __restrict__ g, AggBases bases, int m, const int64_t* __restrict__ seg_off,
                                                   const int64_t* __restrict__ seg_len, const int64_t* __restrict__ src_off,
                                                   const float* __restrict__ coef) {
  const int sgi = blockIdx.y;
  const int64_t off = seg_off[sgi], len = seg_len[sgi];
  const float* c = coef + (size_t)sgi * m;
  const int64_t* so = src_off + (size_t)sgi * m;
  bool vec = agg_al16(g + off);
  for (int j = 0; j < m; ++j)
    if (c[j] != 0.f && so[j] >= 0) vec = vec && agg_al16(bases.p[j] + so[j]);
  const int64_t n4 = vec ? len >> 2 : 0;
  float4* g4 = (float4*)(g + off);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 acc = g4[i];
    for (int j = 0; j < m; ++j) {
      const float cj = c[j];
      if (cj != 0.f && so[j] >= 0) {
        const float4 x = ((const float4*)(bases.p[j] + so[j]))[i];
        acc.x = blend_seq1(acc.x, x.x, cj); acc.y = blend_seq1(acc.y, x.y, cj); acc.z = blend_seq1(acc.z, x.z, cj); acc.w = blend_seq1(acc.w, x.w, cj);
      }
    }
    g4[i] = acc;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
    float acc = g[off + i];
    for (int j = 0; j < m; ++j) {
      const float cj = c[j];
      if (cj != 0.f && so[j] >= 0) acc = blend_seq1(acc, bases.p[j][so[j] + i], cj);
    }
    g[off + i] = acc;
  }
}

extern "C" int fc_aggregate_partial(float* out, const float* global, const float* const* client_bases, int32_t n_clients,
                                    const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* seg_weights,
                                    int32_t n_segments, void* stream) {
  FC_REQUIRE(out && global, "fc_aggregate_partial: null buffer");
  FC_REQUIRE(n_clients >= 0 && n_clients <= FC_AGG_MAX_CLIENTS, "fc_aggregate_partial: %d clients in one call (limit %d)", n_clients, FC_AGG_MAX_CLIENTS);
  FC_REQUIRE(n_clients == 0 || client_bases, "fc_aggregate_partial: null client table");
  if (n_segments <= 0) return 0;
  AggBases b;
  for (int j = 0; j < FC_AGG_MAX_CLIENTS; ++j) b.p[j] = j < n_clients ? client_bases[j] : nullptr;
  hipLaunchKernelGGL(k_blend_v, dim3(AGG_BLOCKS_X, n_segments), dim3(256), 0, (hipStream_t)stream, out, global, b, (int)n_clients, seg_offset, seg_numel,
                     src_offset, seg_weights);
  FC_LAUNCH_CHECK();
  return 0;
}

extern "C" int fc_aggregate(fc_comm_t* comm, float* global, float* partial, int64_t numel, const float* const* client_bases, int32_t n_clients,
                            const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* seg_weights,
                            int32_t n_segments, const int64_t* run_offset, const int64_t* run_numel, int32_t n_runs, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  FC_REQUIRE(global && partial && numel > 0, "fc_aggregate: null buffer");
  FC_REQUIRE(n_clients >= 0 && n_clients <= FC_AGG_MAX_CLIENTS, "fc_aggregate: %d clients in one call (limit %d)", n_clients, FC_AGG_MAX_CLIENTS);
  FC_REQUIRE(n_clients == 0 || client_bases, "fc_aggregate: null client table");
  if (n_segments <= 0) return 0;
  AggBases b;
  for (int j = 0; j < FC_AGG_MAX_CLIENTS; ++j) b.p[j] = j < n_clients ? client_bases[j] : nullptr;
  const bool single = !comm || comm->world == 1;
  // one process: blend straight into the global buffer (every element reads its own old value before writing it)
  float* dst = single ? global : partial;
  hipLaunchKernelGGL(k_blend_v, dim3(AGG_BLOCKS_X, n_segments), dim3(256), 0, s, dst, (const float*)global, b, (int)n_clients, seg_offset, seg_numel, src_offset,
                     seg_weights);
  FC_LAUNCH_CHECK();
  if (single) return 0;
  FC_CHECK_NCCL(rccl()->AllReduce(partial, partial, (size_t)numel, FC_NCCL_FLOAT32, FC_NCCL_SUM, comm->comm, s));
  for (int r = 0; r < n_runs; ++r)      // only the planned (required_params) ranges are replaced
    FC_CHECK_HIP(hipMemcpyAsync(global + run_offset[r], partial + run_offset[r], sizeof(float) * (size_t)run_numel[r], hipMemcpyDeviceToDevice, s));
  return 0;
}

extern "C" int fc_aggregate_blend_seq(float* global, const float* const* client_bases, int32_t n_clients, const int64_t* seg_offset,
                                      const int64_t* seg_numel, const int64_t* src_offset, const float* coef, int32_t n_segments, void* stream) {
  FC_REQUIRE(global, "fc_aggregate_blend_seq: null buffer");
  FC_REQUIRE(n_clients >= 0 && n_clients <= FC_AGG_MAX_CLIENTS, "fc_aggregate_blend_seq: %d clients (limit %d)", n_clients, FC_AGG_MAX_CLIENTS);
  if (n_segments <= 0 || n_clients == 0) return 0;
  AggBases b;
  for (int j = 0; j < FC_AGG_MAX_CLIENTS; ++j) b.p[j] = j < n_clients ? client_bases[j] : nullptr;
  hipLaunchKernelGGL(k_blend_seq, dim3(AGG_BLOCKS_X, n_segments), dim3(256), 0, (hipStream_t)stream, global, b, (int)n_clients, seg_offset, seg_numel, src_offset,
                     coef);
  FC_LAUNCH_CHECK();
  return 0;
}

extern "C" int fc_aggregate_exact(fc_comm_t* comm, float* global, const float* local_client, float* gathered, int64_t slot_numel,
                                  const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* coef,
                                  int32_t n_segments, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  FC_REQUIRE(global && local_client && gathered && slot_numel > 0, "fc_aggregate_exact: null buffer");
  const int world = comm ? comm->world : 1;
  FC_REQUIRE(world <= FC_AGG_MAX_CLIENTS, "fc_aggregate_exact: world %d (limit %d)", world, FC_AGG_MAX_CLIENTS);
  if (!comm) {
    if (gathered != local_client)
      FC_CHECK_HIP(hipMemcpyAsync(gathered, local_client, sizeof(float) * (size_t)slot_numel, hipMemcpyDeviceToDevice, s));
  } else {
    FC_CHECK_NCCL(rccl()->AllGather(local_client, gathered, (size_t)slot_numel, FC_NCCL_FLOAT32, comm->comm, s));
  }
  const float* bases[FC_AGG_MAX_CLIENTS];
  for (int j = 0; j < world; ++j) bases[j] = gathered + (size_t)j * slot_numel;
  return fc_aggregate_blend_seq(global, bases, world, seg_offset, seg_numel, src_offset, coef, n_segments, stream);
}
