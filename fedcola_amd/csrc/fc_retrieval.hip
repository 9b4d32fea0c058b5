// Retrieval evaluation on device: COCOEvaluator.evaluate_recall of the reference
// (src/metrics/eval_coco.py:296-351 with ParallelMatMulModule :48-69).
//
//   sims  = Q . G^T in float64  -- the reference keeps the extracted features in float64 numpy arrays (eval_coco.py:155-156)
//           and multiplies them with Tensor.mm (:55), so the similarity matrix is an fp64 GEMM: v_mfma_f64_16x16x4_f64.
//   ranks = for every query the position, in the descending-similarity order of the gallery, of the best-placed gallery
//           item that carries the query's label (:331-334).  The reference sorts every row and searches it with torch.where
//           per positive; the position of the best positive p* is simply
//               #{g : s_g > s_p*} + #{g < p* : s_g == s_p*}          (ties by ascending gallery index = stable sort)
//           so no sort is needed: one pass finds (s_p*, p*), one pass counts.  Integer outputs, bit-exact.
//
// Queries are processed in batches; the fp64 similarity tile of a batch lives in caller-provided scratch.
#include "../../include/fedcola_hip.h"
#include "fc_kernels.h"

typedef __attribute__((ext_vector_type(4))) double f64x4;

#define RT_BM 64      // queries per block
#define RT_BN 64      // gallery items per block
#define RT_KC 16      // k chunk staged in LDS
#define RT_LD 17      // padded LDS row (doubles)

// S[nq][ng] = Q[nq][d] . G[ng][d]^T (row-major, fp64).  256 threads = 4 waves (2x2), each wave 32x32 = 2x2 MFMA tiles.
// MFMA f64 16x16x4 fragment map: A lane l -> (row l&15, k l>>4); B lane l -> (col l&15, k l>>4);
// C lane l, reg r -> (row 4*r + (l>>4), col l&15)   [verified on gfx950; NOT the f32 16x16x4 map]
__global__ void __launch_bounds__(256) k_sim_f64(const double* __restrict__ Q, const double* __restrict__ G, double* __restrict__ S, int nq, int ng,
                                                 int d) {
  __shared__ double As[RT_BM][RT_LD], Bs[RT_BN][RT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int q0 = blockIdx.y * RT_BM, g0 = blockIdx.x * RT_BN;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
  const int lrow = tid >> 2, lk = (tid & 3) * 4;   // staging: 4 consecutive doubles per thread and operand
  for (int k0 = 0; k0 < d; k0 += RT_KC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + lk + i;
      const int qr = q0 + lrow, gr = g0 + lrow;
      As[lrow][lk + i] = (qr < nq && k < d) ? Q[(size_t)qr * d + k] : 0.0;
      Bs[lrow][lk + i] = (gr < ng && k < d) ? G[(size_t)gr * d + k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < RT_KC; kk += 4) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[wr * 32 + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[wc * 32 + j * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + wr * 32 + i * 16 + 4 * r + (lane >> 4), g = g0 + wc * 32 + j * 16 + (lane & 15);
        if (q < nq && g < ng) S[(size_t)q * ng + g] = acc[i][j][r];
      }
}

// one block per query row of S: best positive (highest similarity, lowest index among equals), then its position
__global__ void __launch_bounds__(256) k_best_rank(const double* __restrict__ S, const int64_t* __restrict__ q_labels,
                                                   const int64_t* __restrict__ g_labels, int ng, int64_t* __restrict__ best) {
  __shared__ double s_val[256];
  __shared__ int s_idx[256];
  __shared__ int s_cnt[256];
  const int q = blockIdx.x, tid = threadIdx.x;
  const double* row = S + (size_t)q * ng;
  const int64_t ql = q_labels[q];
  double bv = 0.0;
  int bi = -1;
  for (int g = tid; g < ng; g += 256) {
    if (g_labels[g] != ql) continue;
    const double v = row[g];
    if (bi < 0 || v > bv) { bv = v; bi = g; }   // ascending g per thread: an equal value never replaces a lower index
  }
  s_val[tid] = bv;
  s_idx[tid] = bi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      const int oi = s_idx[tid + off];
      const double ov = s_val[tid + off];
      const int ci = s_idx[tid];
      const double cv = s_val[tid];
      if (oi >= 0 && (ci < 0 || ov > cv || (ov == cv && oi < ci))) { s_val[tid] = ov; s_idx[tid] = oi; }
    }
    __syncthreads();
  }
  const int p = s_idx[0];
  const double sv = s_val[0];
  if (p < 0) {   // no gallery item carries this label: the reference raises (min() of an empty list); reported as -1
    if (tid == 0) best[q] = -1;
    return;
  }
  int cnt = 0;
  for (int g = tid; g < ng; g += 256) {
    const double v = row[g];
    cnt += (v > sv) || (v == sv && g < p);
  }
  s_cnt[tid] = cnt;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) s_cnt[tid] += s_cnt[tid + off];
    __syncthreads();
  }
  if (tid == 0) best[q] = s_cnt[0];
}

int fc_sim_f64(const double* Q, const double* G, double* S, int nq, int ng, int d, hipStream_t s) {
  if (nq <= 0 || ng <= 0) return 0;
  hipLaunchKernelGGL(k_sim_f64, dim3(fc_cdiv(ng, RT_BN), fc_cdiv(nq, RT_BM)), dim3(256), 0, s, Q, G, S, nq, ng, d);
  FC_LAUNCH_CHECK();
  return 0;
}
int fc_best_rank(const double* S, const int64_t* q_labels, const int64_t* g_labels, int nq, int ng, int64_t* best, hipStream_t s) {
  if (nq <= 0) return 0;
  hipLaunchKernelGGL(k_best_rank, dim3(nq), dim3(256), 0, s, S, q_labels, g_labels, ng, best);
  FC_LAUNCH_CHECK();
  return 0;
}

// queries per batch the scratch can hold (the reference batches 1024 queries, eval_coco.py:298)
extern "C" size_t fc_retrieval_scratch_bytes(int32_t nq_batch, int32_t ng) { return sizeof(double) * (size_t)(nq_batch > 0 ? nq_batch : 0) * (size_t)(ng > 0 ? ng : 0); }

extern "C" int fc_retrieval_best_ranks(const double* q, const double* g, const int64_t* q_labels, const int64_t* g_labels, int32_t nq, int32_t ng,
                                       int32_t d, void* scratch, size_t scratch_bytes, int64_t* best_ranks, void* stream) {
  FC_REQUIRE(nq >= 0 && ng > 0 && d > 0, "retrieval: bad sizes (nq %d, ng %d, d %d)", nq, ng, d);
  FC_REQUIRE(q && g && q_labels && g_labels && best_ranks && scratch, "retrieval: null buffer");
  const size_t row_bytes = sizeof(double) * (size_t)ng;
  const size_t qb = scratch_bytes / row_bytes;
  FC_REQUIRE(qb >= 1, "retrieval: scratch (%zu bytes) holds no similarity row of %d doubles", scratch_bytes, ng);
  hipStream_t s = (hipStream_t)stream;
  for (size_t b0 = 0; b0 < (size_t)nq; b0 += qb) {
    const int nb = (int)((size_t)nq - b0 < qb ? (size_t)nq - b0 : qb);
    FC_TRY(fc_sim_f64(q + b0 * d, g, (double*)scratch, nb, ng, d, s));
    FC_TRY(fc_best_rank((const double*)scratch, q_labels + b0, g_labels, nb, ng, best_ranks + b0, s));
  }
  return 0;
}
// the similarity GEMM alone (unit tests)
extern "C" int fc_k_sim_f64(const double* q, const double* g, double* sims, int32_t nq, int32_t ng, int32_t d, void* stream) {
  FC_REQUIRE(q && g && sims && d > 0, "sim_f64: bad argument");
  return fc_sim_f64(q, g, sims, nq, ng, d, (hipStream_t)stream);
}
