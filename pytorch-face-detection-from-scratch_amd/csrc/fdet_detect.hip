// Detection math of the hot path: target encode, YOLO loss fwd+bwd, decode, greedy NMS,
// step metrics, u8 normalise.  Built with -ffp-contract=off: decode / NMS / IoU results are
// compared BIT-EXACTLY with the reference's CPU arithmetic (separate mul and add, IEEE
// division), so no FMA contraction is allowed in this file.
#include "fdet_common.h"
#include <cfloat>
#include <cmath>

using namespace fdet;

// ======================================================================================
// encode -- datasets/WIDERFace/dataset.py:32-64
// ======================================================================================
__global__ void __launch_bounds__(256)
k_encode(const float* __restrict__ boxes, const int32_t* __restrict__ off, int S,
         double ps_x, double ps_y, float fw, float fh, float* __restrict__ out) {
  const int n = blockIdx.x;
  float* fm = out + (size_t)n * 5 * S * S;
  const int cells = 5 * S * S;
  for (int t = threadIdx.x; t < cells; t += blockDim.x) fm[t] = 0.f;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float psx = (float)ps_x, psy = (float)ps_y;
  const int b0 = off[n], b1 = off[n + 1];
  for (int k = b0; k < b1; ++k) {          // in order: later boxes overwrite (dataset.py:63)
    const float* bx = boxes + (size_t)k * 5;
    const float c = bx[0], x = bx[1], y = bx[2], w = bx[3], h = bx[4];
    float qi = floorf(x / psx), qj = floorf(y / psy);                 // :43
    qi = fminf(fmaxf(qi, -1.0e9f), 1.0e9f);
    qj = fminf(fmaxf(qj, -1.0e9f), 1.0e9f);
    int i = (int)qi, j = (int)qj;
    // :51-52  tensor - python_float(i*ps): the scalar is rounded to fp32 first
    const float ox = (x - (float)((double)i * ps_x)) / psx;          // :55
    const float oy = (y - (float)((double)j * ps_y)) / psy;          // :56
    const float wn = w / fw, hn = h / fh;                             // :58-59
    i = min(max(i, 0), S - 1);                                        // :61
    j = min(max(j, 0), S - 1);                                        // :62
    const int cell = i * S + j;
    fm[0 * S * S + cell] = c;
    fm[1 * S * S + cell] = ox;
    fm[2 * S * S + cell] = oy;
    fm[3 * S * S + cell] = wn;
    fm[4 * S * S + cell] = hn;
  }
}

extern "C" int fdet_encode_targets(const float* boxes, const int32_t* box_offset, int B, int S,
                                   float img_w, float img_h, float* out, void* stream) {
  FDET_REQUIRE(B >= 0 && S > 0 && out && box_offset, "encode: bad arguments (B=%d S=%d)", B, S);
  if (B == 0) return FDET_OK;
  hipLaunchKernelGGL(k_encode, dim3(B), dim3(256), 0, (hipStream_t)stream, boxes, box_offset, S,
                     (double)img_w / S, (double)img_h / S, img_w, img_h, out);
  return check_launch("fdet_encode_targets");
}

// ======================================================================================
// loss fwd + bwd -- losses/YoloLoss.py:4-44, models/ModelMeta.py:173-176
// one wavefront per image; wave-shuffle reduction of the per-cell loss
// ======================================================================================
__global__ void __launch_bounds__(64)
k_yolo_loss(const float* __restrict__ pred, const float* __restrict__ gt, int S,
            float* __restrict__ loss_per_image, float* __restrict__ grad, float grad_scale) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int C = S * S;
  const float* p = pred + (size_t)n * 5 * C;
  const float* g = gt + (size_t)n * 5 * C;
  // :8  torch.nansum(pred) != 0 decides whether nan_to_num runs
  float ns = 0.f;
  for (int t = lane; t < 5 * C; t += 64) { float v = p[t]; ns += (v == v) ? v : 0.f; }
  ns = wave_sum_all(ns);
  const bool fix = (ns != 0.f);           // NaN nansum cannot happen; +-inf != 0 is true
  const float inv_s = (float)(1.0 / (double)S);                       // :25
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    float pv[5], fin[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float v = p[k * C + c];
      fin[k] = 1.f;
      if (fix) {                                                      // :9 nan_to_num(nan=0.1)
        fin[k] = (isfinite(v)) ? 1.f : 0.f;
        if (v != v) v = 0.1f;
        else if (isinf(v)) v = (v > 0.f) ? FLT_MAX : -FLT_MAX;
      }
      pv[k] = v;
    }
    const float g0 = g[c], g1 = g[C + c], g2 = g[2 * C + c], g3 = g[3 * C + c], g4 = g[4 * C + c];
    const float obj = g0, noobj = 1.f - g0;                           // :22-23
    const float dx = g1 - pv[2], dy = g2 - pv[1];                     // :18 pred y,x = ch1,ch2
    const float sg3 = sqrtf(g3), sp3 = sqrtf(pv[3]), sg4 = sqrtf(g4), sp4 = sqrtf(pv[4]);
    const float dw = sg3 - sp3, dh = sg4 - sp4;
    const float cw = 3.f * obj;                                       // :24
    const float xy = cw * (dx * dx + dy * dy);                        // :27-29
    const float wh = cw * (dw * dw + dh * dh);                        // :30-34
    const float wconf = obj + noobj * inv_s;
    const float dc = g0 - pv[0];
    const float conf = wconf * (dc * dc);                             // :36-38
    acc += xy + wh + conf;                                            // :40
    if (grad) {
      float d[5];
      d[0] = wconf * (2.f * dc) * -1.f;
      d[1] = cw * (2.f * dy) * -1.f;
      d[2] = cw * (2.f * dx) * -1.f;
      // autograd of x**0.5: grad * 0.5 * x**(-0.5); 0*inf = NaN is reproduced on purpose (Q8)
      d[3] = (cw * (2.f * dw) * -1.f) * (0.5f * (1.f / sp3));
      d[4] = (cw * (2.f * dh) * -1.f) * (0.5f * (1.f / sp4));
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float v = d[k];
        if (fix) v = v * fin[k];          // nan_to_num backward: grad * isfinite(input)
        grad[(size_t)n * 5 * C + k * C + c] = v * grad_scale;
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) loss_per_image[n] = acc;
}

__global__ void __launch_bounds__(256)
k_sum_fixed(const float* __restrict__ v, int n, float scale, float* __restrict__ out) {
  __shared__ float part[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = part[0] * scale;
}

extern "C" int fdet_yolo_loss_fwd_bwd(const float* pred, const float* gt, int B, int S,
                                      float* loss_per_image, float* loss_sum, float* grad_pred,
                                      float grad_scale, void* stream) {
  FDET_REQUIRE(B > 0 && S > 0 && pred && gt && loss_per_image, "yolo_loss: bad arguments");
  hipLaunchKernelGGL(k_yolo_loss, dim3(B), dim3(64), 0, (hipStream_t)stream, pred, gt, S,
                     loss_per_image, grad_pred, grad_scale);
  if (loss_sum)
    hipLaunchKernelGGL(k_sum_fixed, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_per_image, B, 1.0f,
                       loss_sum);
  return check_launch("fdet_yolo_loss_fwd_bwd");
}

// ======================================================================================
// decode + NMS -- datasets/utils.py:95-170, torchvision.ops.nms 0.11.2 (CPU kernel)
// One workgroup per image; candidates live in LDS.
// ======================================================================================
struct NmsLds {
  float* x1; float* y1; float* x2; float* y2; float* score; float* area;
  int* order;            // sorted position -> candidate index
  unsigned char* dead;   // by sorted position
};

__device__ __forceinline__ NmsLds carve(char* smem, int Kmax) {
  NmsLds L;
  float* f = reinterpret_cast<float*>(smem);
  L.x1 = f; L.y1 = f + Kmax; L.x2 = f + 2 * Kmax; L.y2 = f + 3 * Kmax;
  L.score = f + 4 * Kmax; L.area = f + 5 * Kmax;
  L.order = reinterpret_cast<int*>(f + 6 * Kmax);
  L.dead = reinterpret_cast<unsigned char*>(f + 7 * Kmax);
  return L;
}
__host__ __device__ static inline size_t nms_lds_bytes(int Kmax) { return (size_t)Kmax * 7 * 4 + (size_t)((Kmax + 15) / 16) * 16 + 64; }

// Block-wide ordered compaction of the cells with conf > pt; writes candidates (raw, in
// row-major (i,j) order) into L.* and returns K.  utils.py:111-126,152-155,162.
__device__ int decode_to_lds(const float* __restrict__ m, int S, float pt, float psx, float psy,
                             float fw, float fh, NmsLds L, int* s_base /*LDS [1+nwaves]*/) {
  const int C = S * S;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0) s_base[0] = 0;
  __syncthreads();
  for (int c0 = 0; c0 < C; c0 += blockDim.x) {
    const int c = c0 + tid;
    const bool hit = (c < C) && (m[c] > pt);                          // strict > (:112,:119)
    const unsigned long long bal = __ballot(hit);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_base[1 + wid] = __popcll(bal);
    __syncthreads();
    int base = s_base[0];
    for (int w = 0; w < wid; ++w) base += s_base[1 + w];
    if (hit) {
      const int k = base + before;
      const int i = c / S, j = c - i * S;
      const float X = m[C + c] * psx + (float)i * psx;               // :122 two mults + add
      const float Y = m[2 * C + c] * psy + (float)j * psy;           // :123
      const float Wp = m[3 * C + c] * fw;                            // :124
      const float Hp = m[4 * C + c] * fh;                            // :125
      const float X2 = Wp + X, Y2 = Hp + Y;                          // :153-154
      L.score[k] = m[c];
      L.x1[k] = rintf(X); L.y1[k] = rintf(Y); L.x2[k] = rintf(X2); L.y2[k] = rintf(Y2);   // :162
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < nw; ++w) t += s_base[1 + w]; s_base[0] += t; }
    __syncthreads();
  }
  return s_base[0];
}

// Greedy NMS over the K candidates in L (torchvision 0.11.2 nms_kernel.cpp).  On return
// keep_pos[0..nkeep) holds candidate indices in visiting order (LDS).  Returns nkeep.
__device__ int nms_lds(NmsLds L, int K, double thr, int* keep_idx, int* s_cnt) {
  const int tid = threadIdx.x, nt = blockDim.x;
  // stable descending sort by rank counting: rank = #{j : s_j > s_i or (s_j == s_i and j < i)}
  for (int i = tid; i < K; i += nt) {
    const float si = L.score[i];
    int r = 0;
    for (int j = 0; j < K; ++j) { const float sj = L.score[j]; r += (sj > si) || (sj == si && j < i); }
    L.order[r] = i;
    L.area[i] = (L.x2[i] - L.x1[i]) * (L.y2[i] - L.y1[i]);
  }
  for (int i = tid; i < K; i += nt) L.dead[i] = 0;
  if (tid == 0) s_cnt[0] = 0;
  __syncthreads();
  for (int a = 0; a < K; ++a) {
    if (L.dead[a]) continue;                 // uniform: written before the last barrier
    const int i = L.order[a];
    if (tid == 0) { keep_idx[s_cnt[0]] = i; s_cnt[0] += 1; }
    const float ix1 = L.x1[i], iy1 = L.y1[i], ix2 = L.x2[i], iy2 = L.y2[i], ia = L.area[i];
    for (int b = a + 1 + tid; b < K; b += nt) {
      if (L.dead[b]) continue;
      const int j = L.order[b];
      const float w = fmaxf(0.f, fminf(ix2, L.x2[j]) - fmaxf(ix1, L.x1[j]));
      const float h = fmaxf(0.f, fminf(iy2, L.y2[j]) - fmaxf(iy1, L.y1[j]));
      const float inter = w * h;
      const float ovr = inter / (ia + L.area[j] - inter);           // 0/0 = NaN -> not suppressed
      if ((double)ovr > thr) L.dead[b] = 1;
    }
    __syncthreads();
  }
  __syncthreads();
  return s_cnt[0];
}

__global__ void __launch_bounds__(256)
k_decode(const float* __restrict__ maps, int S, float pt, float psx, float psy, float fw, float fh,
         float* __restrict__ scores, float* __restrict__ boxes, int32_t* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.x, C = S * S;
  NmsLds L = carve(smem, C);
  int* s_base = reinterpret_cast<int*>(smem + nms_lds_bytes(C) - 64);
  const int K = decode_to_lds(maps + (size_t)n * 5 * C, S, pt, psx, psy, fw, fh, L, s_base);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    scores[(size_t)n * C + k] = L.score[k];
    float* b = boxes + ((size_t)n * C + k) * 4;
    b[0] = L.x1[k]; b[1] = L.y1[k]; b[2] = L.x2[k]; b[3] = L.y2[k];
  }
  if (threadIdx.x == 0) counts[n] = K;
}

__global__ void __launch_bounds__(256)
k_nms(const float* __restrict__ boxes, const float* __restrict__ scores, const int32_t* __restrict__ counts,
      int Kmax, double thr, int32_t* __restrict__ keep, int32_t* __restrict__ keep_counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.x;
  NmsLds L = carve(smem, Kmax);
  int* s_cnt = reinterpret_cast<int*>(smem + nms_lds_bytes(Kmax) - 64);
  int* keep_idx = reinterpret_cast<int*>(smem + nms_lds_bytes(Kmax));
  const int K = min(max(counts[n], 0), Kmax);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float* b = boxes + ((size_t)n * Kmax + k) * 4;
    L.x1[k] = b[0]; L.y1[k] = b[1]; L.x2[k] = b[2]; L.y2[k] = b[3];
    L.score[k] = scores[(size_t)n * Kmax + k];
  }
  __syncthreads();
  const int nk = nms_lds(L, K, thr, keep_idx, s_cnt);
  for (int k = threadIdx.x; k < nk; k += blockDim.x) keep[(size_t)n * Kmax + k] = keep_idx[k];
  if (threadIdx.x == 0) keep_counts[n] = nk;
}

__global__ void __launch_bounds__(256)
k_reduce_bbx(const float* __restrict__ maps, int S, float pt, double thr, float psx, float psy, float fw,
             float fh, float* __restrict__ out, int32_t* __restrict__ out_counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.x, C = S * S;
  NmsLds L = carve(smem, C);
  int* s_base = reinterpret_cast<int*>(smem + nms_lds_bytes(C) - 64);
  int* keep_idx = reinterpret_cast<int*>(smem + nms_lds_bytes(C));
  const int K = decode_to_lds(maps + (size_t)n * 5 * C, S, pt, psx, psy, fw, fh, L, s_base);
  __syncthreads();
  const int nk = nms_lds(L, K, thr, keep_idx, s_base);
  for (int k = threadIdx.x; k < nk; k += blockDim.x) {
    const int i = keep_idx[k];
    float* o = out + ((size_t)n * C + k) * 5;
    o[0] = L.score[i]; o[1] = L.x1[i]; o[2] = L.y1[i];
    o[3] = L.x2[i] - L.x1[i];                                        // utils.py:148
    o[4] = L.y2[i] - L.y1[i];                                        // utils.py:149
  }
  if (threadIdx.x == 0) out_counts[n] = nk;
}

template <typename Kern>
static int set_lds(Kern kern, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(FDET_ELAUNCH, "hipFuncSetAttribute(%zu): %s", bytes, hipGetErrorString(e));
  }
  return FDET_OK;
}

extern "C" int fdet_decode(const float* maps, int B, int S, float prob_threshold, float img_w, float img_h,
                           float* scores, float* boxes, int32_t* counts, void* stream) {
  FDET_REQUIRE(B > 0 && S > 0 && S * S <= 4096 && maps && scores && boxes && counts, "decode: bad arguments");
  const size_t lds = nms_lds_bytes(S * S);
  if (int rc = set_lds(k_decode, lds)) return rc;
  const float psx = (float)((double)img_w / S), psy = (float)((double)img_h / S);
  hipLaunchKernelGGL(k_decode, dim3(B), dim3(256), lds, (hipStream_t)stream, maps, S, prob_threshold, psx,
                     psy, img_w, img_h, scores, boxes, counts);
  return check_launch("fdet_decode");
}

extern "C" int fdet_nms(const float* boxes, const float* scores, const int32_t* counts, int B, int Kmax,
                        double iou_threshold, int32_t* keep, int32_t* keep_counts, void* stream) {
  FDET_REQUIRE(B > 0 && Kmax > 0 && Kmax <= 4864 && boxes && scores && counts && keep && keep_counts,
               "nms: bad arguments (Kmax=%d must be in 1..4864: 33 bytes of LDS per candidate)", Kmax);
  const size_t lds = nms_lds_bytes(Kmax) + (size_t)Kmax * 4;
  if (int rc = set_lds(k_nms, lds)) return rc;
  hipLaunchKernelGGL(k_nms, dim3(B), dim3(256), lds, (hipStream_t)stream, boxes, scores, counts, Kmax,
                     iou_threshold, keep, keep_counts);
  return check_launch("fdet_nms");
}

extern "C" int fdet_reduce_bounding_boxes(const float* maps, int B, int S, float prob_threshold,
                                          double iou_threshold, float img_w, float img_h, float* out,
                                          int32_t* out_counts, void* stream) {
  FDET_REQUIRE(B > 0 && S > 0 && S * S <= 4096 && maps && out && out_counts, "reduce_bounding_boxes: bad arguments");
  const size_t lds = nms_lds_bytes(S * S) + (size_t)S * S * 4;
  if (int rc = set_lds(k_reduce_bbx, lds)) return rc;
  const float psx = (float)((double)img_w / S), psy = (float)((double)img_h / S);
  hipLaunchKernelGGL(k_reduce_bbx, dim3(B), dim3(256), lds, (hipStream_t)stream, maps, S, prob_threshold,
                     iou_threshold, psx, psy, img_w, img_h, out, out_counts);
  return check_launch("fdet_reduce_bounding_boxes");
}

// ======================================================================================
// step metrics -- models/ModelMeta.py:199-218 with torchvision.ops.box_iou 0.11.2
// ======================================================================================
__global__ void __launch_bounds__(64)
k_metrics(const float* __restrict__ gt, const int32_t* __restrict__ gtc, const float* __restrict__ pr,
          const int32_t* __restrict__ prc, int Kmax, float* __restrict__ per_image) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int G = gtc[n], P = prc[n];
  float iou_sum = 0.f; int hits = 0;
  if (P > 0) {                                                        // :199
    for (int t = lane; t < G * P; t += 64) {
      const int a = t / P, b = t - a * P;
      const float* ga = gt + ((size_t)n * Kmax + a) * 5;
      const float* pb = pr + ((size_t)n * Kmax + b) * 5;
      const float gx1 = ga[1], gy1 = ga[2], gx2 = ga[3] + ga[1], gy2 = ga[4] + ga[2];   // :201-202
      const float px1 = pb[1], py1 = pb[2], px2 = pb[3] + pb[1], py2 = pb[4] + pb[2];   // :204-205
      const float a1 = (gx2 - gx1) * (gy2 - gy1), a2 = (px2 - px1) * (py2 - py1);
      const float w = fmaxf(fminf(gx2, px2) - fmaxf(gx1, px1), 0.f);
      const float h = fmaxf(fminf(gy2, py2) - fmaxf(gy1, py1), 0.f);
      const float inter = w * h;
      float iou = inter / (a1 + a2 - inter);
      if (iou != iou) iou = 0.f;                                      // nan_to_num(.,0) :206
      else if (isinf(iou)) iou = iou > 0.f ? FLT_MAX : -FLT_MAX;
      hits += (iou > 0.5f);
      iou_sum += iou;
    }
  }
  iou_sum = wave_sum(iou_sum);
  float fh = wave_sum((float)hits);
  if (lane == 0) {
    float recall = 0.f, precision = 0.f;
    if (P > 0) {
      recall = (G == 0) ? 0.f : (float)((double)fh / (double)G);      // :207-210
      precision = (float)((double)fh / (double)P);                    // :212
    }
    per_image[n * 3 + 0] = iou_sum;
    per_image[n * 3 + 1] = recall;
    per_image[n * 3 + 2] = precision;
  }
}

__global__ void __launch_bounds__(64)
k_metrics_total(const float* __restrict__ per_image, int B, float* __restrict__ totals) {
  const int lane = threadIdx.x;
  if (lane < 3) {
    double s = 0.0;
    for (int n = 0; n < B; ++n) s += (double)per_image[n * 3 + lane];
    totals[lane] = (float)(s / (double)B);                            // :216-218
  }
}

extern "C" int fdet_step_metrics(const float* gt, const int32_t* gt_counts, const float* pred,
                                 const int32_t* pred_counts, int B, int Kmax, float* per_image,
                                 float* totals, void* stream) {
  FDET_REQUIRE(B > 0 && Kmax > 0 && gt && pred && gt_counts && pred_counts && per_image, "step_metrics: bad arguments");
  hipLaunchKernelGGL(k_metrics, dim3(B), dim3(64), 0, (hipStream_t)stream, gt, gt_counts, pred, pred_counts,
                     Kmax, per_image);
  if (totals)
    hipLaunchKernelGGL(k_metrics_total, dim3(1), dim3(64), 0, (hipStream_t)stream, per_image, B, totals);
  return check_launch("fdet_step_metrics");
}

// ======================================================================================
// u8 -> f32 / 255 -- models/PoolResnet.py:95, datasets/WIDERFace/dataset.py:146
// ======================================================================================
// x / 255 for an integer 0 <= x <= 255, bit-identical to the IEEE division: q = x * fl(1/255) is off by an ulp for 126 of the
// 256 values, one residual correction (two FMAs) makes all 256 exact (tests/test_gpu_detect.py::test_u8_norm_bit_exact covers every value)
__device__ __forceinline__ float u8_over_255(float x) {
  const float r = 1.0f / 255.0f;
  const float q = x * r;
  const float rem = __builtin_fmaf(-q, 255.0f, x);
  return __builtin_fmaf(rem, r, q);
}

// one dword (4 pixels) in, one float4 out per thread and step: a wave reads 256 contiguous bytes and writes 1 KiB contiguous
__global__ void __launch_bounds__(256)
k_u8_norm(const uint8_t* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t nv = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const unsigned* __restrict__ in4 = reinterpret_cast<const unsigned*>(in);
  float4* __restrict__ out4 = reinterpret_cast<float4*>(out);
  size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; v + 3 * stride < nv; v += 4 * stride) {           // four independent loads in flight
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = in4[v + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 f;
      f.x = u8_over_255((float)(w[k] & 255u));
      f.y = u8_over_255((float)((w[k] >> 8) & 255u));
      f.z = u8_over_255((float)((w[k] >> 16) & 255u));
      f.w = u8_over_255((float)(w[k] >> 24));
      out4[v + k * stride] = f;
    }
  }
  for (; v < nv; v += stride) {
    const unsigned w = in4[v];
    float4 f;
    f.x = u8_over_255((float)(w & 255u));
    f.y = u8_over_255((float)((w >> 8) & 255u));
    f.z = u8_over_255((float)((w >> 16) & 255u));
    f.w = u8_over_255((float)(w >> 24));
    out4[v] = f;
  }
  if (blockIdx.x == 0)
    for (size_t t = nv * 4 + threadIdx.x; t < n; t += blockDim.x) out[t] = u8_over_255((float)in[t]);
}

extern "C" int fdet_u8_to_f32_norm(const uint8_t* in, float* out, size_t n, void* stream) {
  FDET_REQUIRE(in && out, "u8_to_f32_norm: null pointer");
  FDET_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "u8_to_f32_norm: pointers must be 16-byte aligned");
  if (n == 0) return FDET_OK;
  size_t blocks = (n / 16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_u8_norm, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, n);
  return check_launch("fdet_u8_to_f32_norm");
}

// ======================================================================================
// SSD detection math (BASELINE.json config 4: "anchor match + hard-neg mining")
//   multi-scale target encode   datasets/WIDERFace/dataset_ssd.py:36-76,134-139
//   ssd_loss + hard negative mining (+ autograd)   losses/SSDLoss.py:7-86, models/ModelMetaSSD.py:127-129
//   ReduceSSDBoundingBoxes      datasets/utils.py:8-92
// Priors are ordered (scale, i, j); P = sum ps^2 (4774 for 60/30/15/7).
// ======================================================================================
constexpr int SSD_MAXS = 8;
struct SsdScales { int n; int ps[SSD_MAXS]; int start[SSD_MAXS + 1]; };

__global__ void __launch_bounds__(256)
k_ssd_encode(const float* __restrict__ boxes, const int32_t* __restrict__ offs, const SsdScales sc, float fw, float fh,
             float* __restrict__ out) {
  const int n = blockIdx.x, P = sc.start[sc.n];
  float* o = out + (size_t)n * P * 5;
  for (int t = threadIdx.x; t < P * 5; t += blockDim.x) o[t] = 0.f;
  __syncthreads();
  if (threadIdx.x < sc.n) {                                          // one thread per scale: later boxes overwrite
    const int ps = sc.ps[threadIdx.x];
    const float xps = (float)(1.0 / ps);                             // python float 1/ps, used as an fp32 scalar
    const float dconf = (float)(0.001 * ps);
    for (int k = offs[n]; k < offs[n + 1]; ++k) {
      const float* b = boxes + (size_t)k * 5;
      const float xn = b[1] / fw, yn = b[2] / fh, wn = b[3] / fw, hn = b[4] / fh;   // :44-45
      int i = (int)floor((double)(xn / xps)), j = (int)floor((double)(yn / xps));  // :54
      float v1 = xn - (float)((double)i * (1.0 / ps));               // :65
      float v2 = yn - (float)((double)j * (1.0 / ps));
      v1 = v1 / xps; v2 = v2 / xps;                                  // :69-70
      i = min(max(i, 0), ps - 1); j = min(max(j, 0), ps - 1);        // :75-76
      float* d = o + (size_t)(sc.start[threadIdx.x] + i * ps + j) * 5;
      d[0] = b[0] - dconf; d[1] = v1; d[2] = v2; d[3] = wn; d[4] = hn;
    }
  }
}

// one workgroup per image: mining mask, loss partials, un-normalised gradients
__global__ void __launch_bounds__(256)
k_ssd_loss_image(const float* __restrict__ pred, const float* __restrict__ tgt, int P, int ratio,
                 float* __restrict__ grad, unsigned char* __restrict__ mask_out, double* __restrict__ part /*[B][3]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_loss = reinterpret_cast<float*>(smem);                    // -log(conf); -inf for positives
  __shared__ double s_red[3][4];
  __shared__ int s_npos;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* pr = pred + (size_t)n * P * 5;
  const float* tg = tgt + (size_t)n * P * 5;
  if (tid == 0) s_npos = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < P; i += 256) {
    const bool pos = tg[(size_t)i * 5] > 0.f;                        // SSDLoss.py:39
    cnt += pos;
    s_loss[i] = pos ? -INFINITY : -logf(pr[(size_t)i * 5]);          // :45, :68
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if (lane == 0) atomicAdd(&s_npos, cnt);
  __syncthreads();
  const int npos = s_npos;
  const long long nneg = (long long)npos * ratio;                    // :43
  // ---- the nneg largest losses among the negatives (descending sort, ties: lower index first;
  //      :46-52) by radix select on the order-preserving integer image of the floats: four 8-bit
  //      passes find the threshold key T, then keys > T are in and keys == T fill the rest in index order
  const int P_neg = P - npos;
  const long long kk = nneg < (long long)P_neg ? nneg : (long long)P_neg;   // negatives to keep
  __shared__ int s_hist[256];
  __shared__ unsigned s_prefix, s_mask;
  __shared__ int s_want, s_tiebase[5];
  auto key_of = [&](int i) -> unsigned {
    const unsigned b = __float_as_uint(s_loss[i]);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);               // monotone: larger float -> larger key
  };
  if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_want = (int)kk; }
  __syncthreads();
  if (kk > 0 && kk < P_neg) {
    for (int pass = 3; pass >= 0; --pass) {
      for (int t = tid; t < 256; t += 256) s_hist[t] = 0;
      __syncthreads();
      const unsigned prefix = s_prefix, msk = s_mask;
      for (int i = tid; i < P; i += 256) {
        if (s_loss[i] == -INFINITY && tg[(size_t)i * 5] > 0.f) continue;         // positives are not candidates
        const unsigned k = key_of(i);
        if ((k & msk) == prefix) atomicAdd(&s_hist[(k >> (8 * pass)) & 255u], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int want = s_want, b = 255;
        for (; b > 0; --b) { if (s_hist[b] >= want) break; want -= s_hist[b]; }
        s_want = want;                                               // rank inside the chosen bin (1-based count still to take)
        s_prefix = prefix | ((unsigned)b << (8 * pass));
        s_mask = msk | (0xFFu << (8 * pass));
      }
      __syncthreads();
    }
  }
  const unsigned Tkey = s_prefix;                                    // threshold key (valid when 0 < kk < P_neg)
  const int ties_to_take = s_want;                                   // how many keys == T are kept (lowest indices)
  __syncthreads();
  double bce = 0.0, sl1 = 0.0;
  int tie_seen_base = 0;                                             // ties with a lower index in earlier sweeps
  for (int i0 = 0; i0 < P; i0 += 256) {
    const int i = i0 + tid;
    const bool inb = i < P;
    const float label = inb ? tg[(size_t)i * 5] : 0.f;
    const bool pos = inb && label > 0.f;
    bool sel = pos;
    bool tie = false;
    if (inb && !pos) {
      if (kk >= P_neg) sel = kk > 0;
      else if (kk > 0) { const unsigned k = key_of(i); sel = k > Tkey; tie = k == Tkey; }
    }
    // ordered count of the ties in this sweep (index order = lane order inside the sweep)
    const unsigned long long bal = __ballot(tie);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_tiebase[1 + wid] = __popcll(bal);
    __syncthreads();
    int base = tie_seen_base;
    for (int w = 0; w < wid; ++w) base += s_tiebase[1 + w];
    if (tie) sel = base + before < ties_to_take;
    tie_seen_base += s_tiebase[1] + s_tiebase[2] + s_tiebase[3] + s_tiebase[4];
    __syncthreads();
    if (!inb) continue;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
    if (sel) {
      const float c0 = pr[(size_t)i * 5];
      const float lo = 1e-7f, hi = 1.f - 1e-7f;                      // 10**-7 as an fp32 scalar (:13-14)
      const float c = fminf(fmaxf(c0, lo), hi);
      const float lr = rintf(label);                                 // :72
      bce += (double)(-1.f * (lr * logf(c) + (1.f - lr) * logf(1.f - c)));
      if (c0 >= lo && c0 <= hi) g0 = -(lr / c) + (1.f - lr) / (1.f - c);
    }
    if (pos) {
      const float* pl = pr + (size_t)i * 5 + 1;
      const float* gl = tg + (size_t)i * 5 + 1;
      float gg[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = pl[k] - gl[k], ad = fabsf(d);
        sl1 += ad < 1.f ? (double)(0.5f * d * d) : (double)(ad - 0.5f);           // F.smooth_l1_loss, beta = 1
        gg[k] = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
      }
      g1 = gg[0]; g2 = gg[1]; g3 = gg[2]; g4 = gg[3];
    }
    if (grad) {
      float* go = grad + ((size_t)n * P + i) * 5;
      go[0] = g0; go[1] = g1; go[2] = g2; go[3] = g3; go[4] = g4;
    }
    if (mask_out) mask_out[(size_t)n * P + i] = sel ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) { bce += __shfl_down(bce, off, 64); sl1 += __shfl_down(sl1, off, 64); }
  if (lane == 0) { s_red[0][wid] = bce; s_red[1][wid] = sl1; }
  __syncthreads();
  if (tid == 0) {
    part[(size_t)n * 3 + 0] = ((s_red[0][0] + s_red[0][1]) + s_red[0][2]) + s_red[0][3];
    part[(size_t)n * 3 + 1] = ((s_red[1][0] + s_red[1][1]) + s_red[1][2]) + s_red[1][3];
    part[(size_t)n * 3 + 2] = (double)npos;
  }
}

__global__ void __launch_bounds__(64)
k_ssd_loss_total(const double* __restrict__ part, int B, float* __restrict__ loss, float* __restrict__ inv_npos) {
  if (threadIdx.x == 0) {
    double bce = 0.0, sl1 = 0.0, np = 0.0;
    for (int n = 0; n < B; ++n) { bce += part[n * 3]; sl1 += part[n * 3 + 1]; np += part[n * 3 + 2]; }
    loss[0] = (float)((sl1 + bce) / np);                             // :86 (0/0 = NaN as in the reference)
    inv_npos[0] = (float)(1.0 / np);
  }
}

// data-parallel form: the three batch sums (BCE, smooth-L1, positive priors) of THIS shard, fixed order, fp64
__global__ void __launch_bounds__(64)
k_ssd_loss_sums(const double* __restrict__ part, int B, double* __restrict__ sums) {
  if (threadIdx.x == 0) {
    double bce = 0.0, sl1 = 0.0, np = 0.0;
    for (int n = 0; n < B; ++n) { bce += part[n * 3]; sl1 += part[n * 3 + 1]; np += part[n * 3 + 2]; }
    sums[0] = bce; sums[1] = sl1; sums[2] = np;
  }
}
__global__ void __launch_bounds__(64)
k_ssd_loss_finish(const double* __restrict__ sums, float* __restrict__ loss, float* __restrict__ inv_npos) {
  if (threadIdx.x == 0) {
    loss[0] = (float)((sums[1] + sums[0]) / sums[2]);                  // :86 with the batch-wide sums
    inv_npos[0] = (float)(1.0 / sums[2]);
  }
}

__global__ void __launch_bounds__(256)
k_ssd_scale_grad(float* __restrict__ grad, size_t n, const float* __restrict__ inv_npos) {
  const float s = inv_npos[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) grad[i] *= s;
}

// one workgroup per image: decode with priors -> threshold -> round -> greedy NMS -> xywh
__global__ void __launch_bounds__(256)
k_ssd_reduce(const float* __restrict__ x, const SsdScales sc, int with_priors, const float* __restrict__ priors, float pt,
             double thr, float fw, float fh, float* __restrict__ out, int32_t* __restrict__ out_counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.x, P = sc.start[sc.n];
  NmsLds L = carve(smem, P);
  int* s_base = reinterpret_cast<int*>(smem + nms_lds_bytes(P) - 64);
  int* keep_idx = reinterpret_cast<int*>(smem + nms_lds_bytes(P));
  const float* m = x + (size_t)n * P * 5;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0) s_base[0] = 0;
  __syncthreads();
  for (int c0 = 0; c0 < P; c0 += blockDim.x) {                        // ordered compaction (utils.py:54-59)
    const int c = c0 + tid;
    const bool hit = (c < P) && (m[(size_t)c * 5] > pt);
    const unsigned long long bal = __ballot(hit);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_base[1 + wid] = __popcll(bal);
    __syncthreads();
    int base = s_base[0];
    for (int w = 0; w < wid; ++w) base += s_base[1 + w];
    if (hit) {
      const int k = base + before;
      int s_ = 0;
      while (s_ + 1 < sc.n && c >= sc.start[s_ + 1]) ++s_;
      const int ps = sc.ps[s_], loc = c - sc.start[s_];
      const int i = loc / ps, j = loc - i * ps;
      const float* r = m + (size_t)c * 5;
      float v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4];
      if (with_priors) {                                             // :63-68
        const float mult = (float)(1.0 / ps);
        v1 = v1 * mult; v2 = v2 * mult;
        if (priors) {                                                // a caller-supplied (P,4) prior table (:31-32): added to all of x, y, w, h
          const float* pr = priors + (size_t)c * 4;
          v1 = v1 + pr[0]; v2 = v2 + pr[1]; v3 = v3 + pr[2]; v4 = v4 + pr[3];
        } else {                                                     // calculate_priors (:36-48): (i / ps, j / ps, 0, 0)
          v1 = v1 + (float)i * mult; v2 = v2 + (float)j * mult;
        }
      }
      const float X = v1 * fw, Y = v2 * fh, Wd = v3 * fw, Hd = v4 * fh;           // :69-70 (width, height as named there)
      const float X2 = Wd + X, Y2 = Hd + Y;                          // :79-80
      L.score[k] = r[0];
      L.x1[k] = rintf(X); L.y1[k] = rintf(Y); L.x2[k] = rintf(X2); L.y2[k] = rintf(Y2);   // :84
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < nw; ++w) t += s_base[1 + w]; s_base[0] += t; }
    __syncthreads();
  }
  const int K = s_base[0];
  __syncthreads();
  const int nk = nms_lds(L, K, thr, keep_idx, s_base);
  for (int k = tid; k < nk; k += blockDim.x) {
    const int i = keep_idx[k];
    float* o = out + ((size_t)n * P + k) * 5;
    o[0] = L.score[i]; o[1] = L.x1[i]; o[2] = L.y1[i];
    o[3] = L.x2[i] - L.x1[i];                                        // :73-74
    o[4] = L.y2[i] - L.y1[i];
  }
  if (tid == 0) out_counts[n] = nk;
}

static int ssd_scales(const int* h_ps, int ns, SsdScales& sc) {
  if (!h_ps || ns < 1 || ns > SSD_MAXS) return fail(FDET_EINVAL, "ssd: 1..%d patch sizes (got %d)", SSD_MAXS, ns);
  sc.n = ns; sc.start[0] = 0;
  for (int s = 0; s < ns; ++s) {
    if (h_ps[s] < 1 || h_ps[s] > 1024) return fail(FDET_EINVAL, "ssd: bad patch size %d", h_ps[s]);
    sc.ps[s] = h_ps[s]; sc.start[s + 1] = sc.start[s] + h_ps[s] * h_ps[s];
  }
  return FDET_OK;
}

extern "C" int fdet_ssd_num_priors(const int* h_patch_sizes, int nscales) {
  SsdScales sc{};
  if (ssd_scales(h_patch_sizes, nscales, sc)) return -1;
  return sc.start[sc.n];
}

extern "C" int fdet_ssd_encode_targets(const float* boxes, const int32_t* box_offsets, int B, const int* h_patch_sizes,
                                       int nscales, float img_w, float img_h, float* out, void* stream) {
  FDET_REQUIRE(boxes && box_offsets && out && B > 0 && img_w > 0 && img_h > 0, "ssd_encode_targets: bad arguments");
  SsdScales sc{};
  if (int rc = ssd_scales(h_patch_sizes, nscales, sc)) return rc;
  hipLaunchKernelGGL(k_ssd_encode, dim3(B), dim3(256), 0, (hipStream_t)stream, boxes, box_offsets, sc, img_w, img_h, out);
  return check_launch("fdet_ssd_encode_targets");
}

extern "C" size_t fdet_ssd_loss_ws_bytes(int B) { return B > 0 ? (size_t)B * 3 * sizeof(double) + 16 : 0; }

extern "C" int fdet_ssd_loss_fwd_bwd(const float* pred, const float* target, int B, int P, int neg_pos_ratio, float* loss,
                                     float* grad, uint8_t* mask, void* ws, size_t ws_bytes, void* stream) {
  FDET_REQUIRE(pred && target && loss && ws && B > 0 && P > 0 && neg_pos_ratio >= 0, "ssd_loss_fwd_bwd: bad arguments");
  FDET_REQUIRE((size_t)P * 4 <= 150 * 1024, "ssd_loss_fwd_bwd: P=%d priors exceed the LDS tile", P);
  if (ws_bytes < fdet_ssd_loss_ws_bytes(B)) return fail(FDET_EWORKSPACE, "ssd_loss_fwd_bwd: workspace %zu < %zu bytes", ws_bytes, fdet_ssd_loss_ws_bytes(B));
  double* part = (double*)ws;
  float* inv = (float*)(part + (size_t)B * 3);
  const size_t lds = (size_t)P * 4;
  if (int rc = set_lds(k_ssd_loss_image, lds)) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ssd_loss_image, dim3(B), dim3(256), lds, st, pred, target, P, neg_pos_ratio, grad, mask, part);
  if (int rc = check_launch("fdet_ssd_loss_fwd_bwd")) return rc;
  hipLaunchKernelGGL(k_ssd_loss_total, dim3(1), dim3(64), 0, st, part, B, loss, inv);
  if (grad) {
    const size_t n = (size_t)B * P * 5;
    size_t blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_ssd_scale_grad, dim3((unsigned)blocks), dim3(256), 0, st, grad, n, inv);
  }
  return check_launch("fdet_ssd_loss_fwd_bwd(total)");
}

extern "C" int fdet_ssd_loss_parts(const float* pred, const float* target, int B, int P, int neg_pos_ratio, float* grad,
                                   uint8_t* mask, double* sums, void* ws, size_t ws_bytes, void* stream) {
  FDET_REQUIRE(pred && target && sums && ws && B > 0 && P > 0 && neg_pos_ratio >= 0, "ssd_loss_parts: bad arguments");
  FDET_REQUIRE((size_t)P * 4 <= 150 * 1024, "ssd_loss_parts: P=%d priors exceed the LDS tile", P);
  if (ws_bytes < fdet_ssd_loss_ws_bytes(B)) return fail(FDET_EWORKSPACE, "ssd_loss_parts: workspace %zu < %zu bytes", ws_bytes, fdet_ssd_loss_ws_bytes(B));
  double* part = (double*)ws;
  const size_t lds = (size_t)P * 4;
  if (int rc = set_lds(k_ssd_loss_image, lds)) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ssd_loss_image, dim3(B), dim3(256), lds, st, pred, target, P, neg_pos_ratio, grad, mask, part);
  if (int rc = check_launch("fdet_ssd_loss_parts")) return rc;
  hipLaunchKernelGGL(k_ssd_loss_sums, dim3(1), dim3(64), 0, st, part, B, sums);
  return check_launch("fdet_ssd_loss_parts(sums)");
}

extern "C" int fdet_ssd_loss_finish(const double* sums, float* loss, float* grad, size_t n_grad, void* ws, size_t ws_bytes,
                                    void* stream) {
  FDET_REQUIRE(sums && loss && ws && ws_bytes >= 16, "ssd_loss_finish: bad arguments (workspace of at least 16 bytes)");
  float* inv = (float*)ws;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ssd_loss_finish, dim3(1), dim3(64), 0, st, sums, loss, inv);
  if (grad && n_grad) {
    size_t blocks = (n_grad + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_ssd_scale_grad, dim3((unsigned)blocks), dim3(256), 0, st, grad, n_grad, inv);
  }
  return check_launch("fdet_ssd_loss_finish");
}

// priors: NULL = the reference's calculate_priors(); else a device (P,4) table (ReduceSSDBoundingBoxes(priors=...), utils.py:31-32)
extern "C" int fdet_ssd_reduce_bounding_boxes_priors(const float* x, int B, const int* h_patch_sizes, int nscales, int with_priors,
                                                     const float* priors, float prob_threshold, double iou_threshold, float img_w,
                                                     float img_h, float* out, int32_t* out_counts, void* stream);
extern "C" int fdet_ssd_reduce_bounding_boxes(const float* x, int B, const int* h_patch_sizes, int nscales, int with_priors,
                                              float prob_threshold, double iou_threshold, float img_w, float img_h,
                                              float* out, int32_t* out_counts, void* stream) {
  return fdet_ssd_reduce_bounding_boxes_priors(x, B, h_patch_sizes, nscales, with_priors, nullptr, prob_threshold, iou_threshold, img_w,
                                               img_h, out, out_counts, stream);
}
extern "C" int fdet_ssd_reduce_bounding_boxes_priors(const float* x, int B, const int* h_patch_sizes, int nscales, int with_priors,
                                                     const float* priors, float prob_threshold, double iou_threshold, float img_w,
                                                     float img_h, float* out, int32_t* out_counts, void* stream) {
  FDET_REQUIRE(x && out && out_counts && B > 0, "ssd_reduce_bounding_boxes: bad arguments");
  SsdScales sc{};
  if (int rc = ssd_scales(h_patch_sizes, nscales, sc)) return rc;
  const int P = sc.start[sc.n];
  const size_t lds = nms_lds_bytes(P) + (size_t)P * 4;
  FDET_REQUIRE(lds <= 160 * 1024, "ssd_reduce_bounding_boxes: %d priors need %zu bytes of LDS (> 160 KB)", P, lds);
  if (int rc = set_lds(k_ssd_reduce, lds)) return rc;
  hipLaunchKernelGGL(k_ssd_reduce, dim3(B), dim3(256), lds, (hipStream_t)stream, x, sc, with_priors, priors, prob_threshold,
                     iou_threshold, img_w, img_h, out, out_counts);
  return check_launch("fdet_ssd_reduce_bounding_boxes");
}

// SSD heads: z [N,CP,ps,ps] (channels 0..4 = Linear(C,5) on the NHWC map, models/SSD.py:233-238) ->
// rows prior_start + i*ps + j of y [N,P,5]: sigmoid on the score (:240), apply_priors (:206-218).
__global__ void __launch_bounds__(256)
k_ssd_head_pack_fwd(const float* __restrict__ z, int N, int CP, int ps, int prior_start, int P, float* __restrict__ y) {
  const int cells = ps * ps;
  const float mult = (float)(1.0 / ps);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < N * cells; t += gridDim.x * 256) {
    const int n = t / cells, c = t - n * cells;
    const int i = c / ps, j = c - i * ps;
    const float* zp = z + (size_t)n * CP * cells + c;
    float* o = y + ((size_t)n * P + prior_start + c) * 5;
    o[0] = 1.f / (1.f + expf(-zp[0]));
    o[1] = zp[cells] * mult + (float)i * mult;
    o[2] = zp[2 * cells] * mult + (float)j * mult;
    o[3] = zp[3 * cells];
    o[4] = zp[4 * cells];
  }
}
__global__ void __launch_bounds__(256)
k_ssd_head_pack_bwd(const float* __restrict__ dy, const float* __restrict__ y, int N, int CP, int ps, int prior_start,
                    int P, float* __restrict__ dz) {
  const int cells = ps * ps;
  const float mult = (float)(1.0 / ps);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < N * cells; t += gridDim.x * 256) {
    const int n = t / cells, c = t - n * cells;
    const float* g = dy + ((size_t)n * P + prior_start + c) * 5;
    const float s = y[((size_t)n * P + prior_start + c) * 5];
    float* d = dz + (size_t)n * CP * cells + c;
    d[0] = g[0] * (s * (1.f - s));
    d[cells] = g[1] * mult;
    d[2 * cells] = g[2] * mult;
    d[3 * cells] = g[3];
    d[4 * cells] = g[4];
    for (int k = 5; k < CP; ++k) d[(size_t)k * cells] = 0.f;
  }
}

extern "C" int fdet_ssd_head_pack_fwd(const float* z, int N, int CP, int ps, int prior_start, int P, float* y, void* stream) {
  FDET_REQUIRE(z && y && N > 0 && CP >= 5 && ps > 0 && prior_start >= 0 && prior_start + ps * ps <= P, "ssd_head_pack_fwd: bad arguments");
  const int total = N * ps * ps;
  hipLaunchKernelGGL(k_ssd_head_pack_fwd, dim3(min((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, z, N, CP, ps, prior_start, P, y);
  return check_launch("fdet_ssd_head_pack_fwd");
}
extern "C" int fdet_ssd_head_pack_bwd(const float* dy, const float* y, int N, int CP, int ps, int prior_start, int P, float* dz,
                                      void* stream) {
  FDET_REQUIRE(dy && y && dz && N > 0 && CP >= 5 && ps > 0 && prior_start >= 0 && prior_start + ps * ps <= P, "ssd_head_pack_bwd: bad arguments");
  const int total = N * ps * ps;
  hipLaunchKernelGGL(k_ssd_head_pack_bwd, dim3(min((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, dy, y, N, CP, ps, prior_start, P, dz);
  return check_launch("fdet_ssd_head_pack_bwd");
}

extern "C" int fdet_version(void) { return FDET_VERSION; }
extern "C" const char* fdet_last_error(void) { return fdet::err_buf(); }
