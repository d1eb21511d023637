// Detection kernels: batched greedy NMS (64-wide bitmask rows: one wavefront lane per
// mask bit), pairwise IoU, the fused glue kernels (RoIAlign: roi_align.hip).  fp32 box maths with FMA
// contraction disabled so the keep decisions match the scalar CPU oracle bit for bit.
#include "hd_common.h"
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float box_iou_dev(const float* a, const float* b) {
  float area_a = (a[2] - a[0]) * (a[3] - a[1]);
  float area_b = (b[2] - b[0]) * (b[3] - b[1]);
  float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
  float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
  float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  float inter = w * h;
  return inter / (area_a + area_b - inter);
}

// grid (colblk, rowblk, B), block 64.  mask[b][i][colblk] bit j set <=> IoU(box i, box col*64+j) > thr, j-index > i
__global__ void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts, int nmax, float thr,
                                uint64_t* __restrict__ mask) {
  const int b = blockIdx.z;
  const int n = counts[b];
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  if (row0 >= n || col0 >= n || blockIdx.x < blockIdx.y) return;
  const int cb = (nmax + 63) / 64;
  __shared__ float cbx[64 * 4];
  const float* bb = boxes + (size_t)b * nmax * 4;
  int t = threadIdx.x;
  int ncol = min(64, n - col0);
  if (t < ncol) {
    cbx[t * 4 + 0] = bb[(col0 + t) * 4 + 0];
    cbx[t * 4 + 1] = bb[(col0 + t) * 4 + 1];
    cbx[t * 4 + 2] = bb[(col0 + t) * 4 + 2];
    cbx[t * 4 + 3] = bb[(col0 + t) * 4 + 3];
  }
  __syncthreads();
  int i = row0 + t;
  if (i < n) {
    float me[4] = {bb[i * 4], bb[i * 4 + 1], bb[i * 4 + 2], bb[i * 4 + 3]};
    uint64_t m = 0;
    int start = (row0 == col0) ? t + 1 : 0;
    for (int j = start; j < ncol; ++j)
      if (box_iou_dev(me, cbx + j * 4) > thr) m |= (1ull << j);
    mask[((size_t)b * nmax + i) * cb + blockIdx.x] = m;
  }
}

// one 256-thread block per image.  Wave 0 resolves each 64-box chunk serially (ALU only); all four waves then OR the
// mask rows of the kept boxes into the "removed" words -- 16 row loads in flight per wave, every wave a quarter of the
// rows -- and the partial words are combined through LDS.  n <= 16384.
// `pick` (optional): the candidate index order[b][i] of the kept boxes, in order, for the first pick_n of them (rest: order[b][0]), and
// picked[b] = how many -- the compaction callers otherwise build from `keep` with a prefix sum and a scatter.
__global__ __launch_bounds__(256) void nms_reduce_kernel(const uint64_t* __restrict__ mask, const int* __restrict__ counts, int nmax,
                                                         uint8_t* __restrict__ keep, int max_keep, const int64_t* __restrict__ order,
                                                         int64_t* __restrict__ pick, int pick_n, int64_t* __restrict__ picked) {
  constexpr int MAXW = 4;  // 64*64*4 boxes
  __shared__ uint64_t remv[MAXW * 64];
  __shared__ uint64_t part[4][MAXW * 64];
  __shared__ uint64_t s_keepbits;
  __shared__ int s_kept;
  const int b = blockIdx.x;
  const int n = counts[b];
  const int cb = (nmax + 63) / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t* mb = mask + (size_t)b * nmax * cb;
  uint8_t* kb = keep + (size_t)b * nmax;
  const int nchunks = (n + 63) / 64;
  for (int i = tid; i < MAXW * 64; i += 256) remv[i] = 0;
  if (tid == 0) s_kept = 0;
  if (pick)
    for (int i = tid; i < pick_n; i += 256) pick[(size_t)b * pick_n + i] = order[(size_t)b * nmax];   // padding = the first candidate, as gather(order, 0) gives
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    // callers that only use the first max_keep survivors (post-NMS top-n): once that many are kept, nothing later can be
    // selected -- mark the rest not kept and stop the serial scan
    if (s_kept >= max_keep) {
      for (int i = c * 64 + tid; i < n; i += 256) kb[i] = 0;
      break;
    }
    const int lim = min(64, n - c * 64);
    if (wave == 0) {
      const int i = c * 64 + lane;
      const uint64_t diag = (i < n) ? mb[(size_t)i * cb + c] : 0;
      // greedy resolve of the chunk on the SCALAR unit: `alive` is wave-uniform, only the boxes still alive are visited, and box j's
      // diagonal word comes from lane j by v_readlane (the __shfl form of this loop compiled to two ds_bpermute per step: ~6 us per
      // 64-box chunk, the whole cost of the scan)
      const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
      uint64_t alive = ~remv[c];
      if (lim < 64) alive &= (1ull << lim) - 1ull;
      alive = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(alive >> 32)) << 32) |
              (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)alive);
      uint64_t keepbits = 0;
      for (int it = 0; it < 64 && alive; ++it) {          // (bounded: at most one visit per box)
        const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));
        keepbits |= (1ull << j);
        const uint64_t dj = ((uint64_t)(unsigned)__builtin_amdgcn_readlane(dhi, j) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readlane(dlo, j);
        alive &= ~dj;
        alive &= ~(1ull << j);
      }
      if (i < n) kb[i] = (uint8_t)((keepbits >> lane) & 1ull);
      if (pick && ((keepbits >> lane) & 1ull)) {
        const int pos = s_kept + __popcll(keepbits & ((1ull << lane) - 1ull));
        if (pos < pick_n) pick[(size_t)b * pick_n + pos] = order[(size_t)b * nmax + i];
      }
      if (lane == 0) {
        s_keepbits = keepbits;
        s_kept += __popcll(keepbits);
      }
    }
    __syncthreads();
    const uint64_t keepbits = s_keepbits;
    // words beyond this chunk: w in (c, nchunks)
    const int nw = nchunks - (c + 1);
    if (nw > 0) {
      for (int wb = 0; wb < nw; wb += 64) {
        const int w = c + 1 + wb + lane;
        const bool act = w < nchunks;
        uint64_t accw = 0;
        // this wave's quarter of the rows, 16 loads in flight
        for (int jb = wave * 16; jb < lim; jb += 64) {
          uint64_t rws[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int j = min(jb + u, lim - 1);
            rws[u] = act ? mb[(size_t)(c * 64 + j) * cb + w] : 0ull;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (jb + u < lim && ((keepbits >> (jb + u)) & 1ull)) accw |= rws[u];
        }
        part[wave][wb + lane] = accw;
      }
      __syncthreads();
      for (int i = tid; i < nw; i += 256) remv[c + 1 + i] |= part[0][i] | part[1][i] | part[2][i] | part[3][i];
    }
    __syncthreads();
  }
  if (picked && tid == 0) picked[b] = min(s_kept, pick_n);
}

// sorted, category-shifted boxes for batched NMS (torchvision _batched_nms_coordinate_trick [EXT]): per image
// counts = #valid, bmax = max coordinate over the valid boxes (0 when there is none),
// sorted[j] = boxes[order[j]] + idxs[order[j]] * (bmax + 1)
__global__ __launch_bounds__(256) void nms_prepare_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ idxs,
                                                          const uint8_t* __restrict__ valid, const int64_t* __restrict__ order, int n,
                                                          float* __restrict__ sorted, int* __restrict__ counts) {
  __shared__ float s_max[4];
  __shared__ int s_cnt[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* bb = boxes + (size_t)b * n * 4;
  const uint8_t* vb = valid + (size_t)b * n;
  float m = -INFINITY;
  int cnt = 0;
  for (int i = tid; i < n; i += 256)
    if (vb[i]) {
      const float4 v = *reinterpret_cast<const float4*>(bb + (size_t)i * 4);
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      ++cnt;
    }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m = fmaxf(m, __shfl_xor(m, d));
    cnt += __shfl_xor(cnt, d);
  }
  if ((tid & 63) == 0) {
    s_max[tid >> 6] = m;
    s_cnt[tid >> 6] = cnt;
  }
  __syncthreads();
  cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  const float scale = (cnt > 0 ? m : 0.f) + 1.f;
  if (tid == 0) counts[b] = cnt;
  for (int j = tid; j < n; j += 256) {
    const int64_t c = order[(size_t)b * n + j];
    const float off = (float)idxs[(size_t)b * n + c] * scale;
    float4 v = *reinterpret_cast<const float4*>(bb + (size_t)c * 4);
    v.x += off; v.y += off; v.z += off; v.w += off;
    *reinterpret_cast<float4*>(sorted + ((size_t)b * n + j) * 4) = v;
  }
}

// ---- batched NMS over SEGMENTS (hd_batched_nms_pick_segments).  torchvision's batched_nms shifts the boxes of category c by
// c * (max coordinate + 1): boxes of different categories never suppress each other, so the greedy scan decomposes into one
// independent scan per (image, category).  The RPN's candidates arrive level by level (<= 1000 per level, already in descending
// score order inside a level: rpn._get_top_n_idx): 24 x 5 scans of <= 16 chunks instead of 24 scans of 53 chunks whose blocks
// occupied 24 of 256 CUs for 200 us, and a fifth of the pair tests.  The shift is still applied (per-image maximum, as
// torchvision computes it), so every IoU is the bit pattern the one-list form produces.
struct NmsSegs {
  int off[9];      // segment l = candidates [off[l], off[l+1]) of every row
  int L, S;        // number of segments (<= 8), largest segment (the per-segment stride of the work arrays)
};

// one block per image: per-image maximum coordinate, then per segment the valid candidates, in their given order, compacted to
// the front of the segment's slot: shifted boxes, candidate indices, count
__global__ __launch_bounds__(256) void nms_seg_prepare_kernel(const float* __restrict__ boxes, const uint8_t* __restrict__ valid, int n, NmsSegs sg,
                                                              float* __restrict__ sorted, int64_t* __restrict__ order, int* __restrict__ counts) {
  __shared__ float s_max[4];
  __shared__ int s_cnt[4];
  __shared__ int s_wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* bb = boxes + (size_t)b * n * 4;
  const uint8_t* vb = valid + (size_t)b * n;
  float m = -INFINITY;
  int cnt = 0;
  for (int i = tid; i < n; i += 256)
    if (vb[i]) {
      const float4 v = *reinterpret_cast<const float4*>(bb + (size_t)i * 4);
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      ++cnt;
    }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m = fmaxf(m, __shfl_xor(m, d));
    cnt += __shfl_xor(cnt, d);
  }
  if (lane == 0) {
    s_max[wave] = m;
    s_cnt[wave] = cnt;
  }
  __syncthreads();
  cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  const float scale = (cnt > 0 ? m : 0.f) + 1.f;
  for (int l = 0; l < sg.L; ++l) {
    const int lo = sg.off[l], hi = sg.off[l + 1];
    const float off = (float)l * scale;
    float* so = sorted + ((size_t)b * sg.L + l) * sg.S * 4;
    int64_t* oo = order + ((size_t)b * sg.L + l) * sg.S;
    int base = 0;
    for (int c0 = lo; c0 < hi; c0 += 256) {
      const int i = c0 + tid;
      const bool v = i < hi && vb[i];
      const uint64_t bal = __ballot(v);
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      __syncthreads();                       // s_wsum of the previous round has been read
      if (lane == 0) s_wsum[wave] = __popcll(bal);
      __syncthreads();
      int wbase = 0;
      for (int w = 0; w < wave; ++w) wbase += s_wsum[w];
      if (v) {
        const int r = base + wbase + before;
        float4 bx = *reinterpret_cast<const float4*>(bb + (size_t)i * 4);
        bx.x += off; bx.y += off; bx.z += off; bx.w += off;
        *reinterpret_cast<float4*>(so + (size_t)r * 4) = bx;
        oo[r] = i;
      }
      base += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    }
    if (tid == 0) counts[b * sg.L + l] = base;
  }
}

// one block per (image, segment): merge of the segments' survivor lists (each in descending score order) into the row's global order --
// descending score, equal scores by ascending candidate index, i.e. the stable sort the one-list form starts from.  The global rank of a
// survivor = its rank in its own list + for every other segment the number of ITS survivors that sort before it (binary search over
// that segment's scores, staged in LDS; lower segments hold lower candidate indices, so they win ties).  pick[b][rank] = candidate;
// ranks >= top_n are dropped; positions past the total are padding = the best candidate (as hd_batched_nms_pick pads); picked[b] = total.
__global__ __launch_bounds__(256) void nms_seg_merge_kernel(const float* __restrict__ scores, const int64_t* __restrict__ pick_seg,
                                                            const int64_t* __restrict__ picked_seg, int n, int pick_n_seg, NmsSegs sg, int top_k,
                                                            int64_t* __restrict__ pick, int64_t* __restrict__ picked) {
  extern __shared__ float s_sc[];                 // [L][pick_n_seg] survivor scores of every segment of this row
  const int b = blockIdx.x / sg.L, l = blockIdx.x % sg.L;
  const float* sb = scores + (size_t)b * n;
  int cnt[8], total = 0;
  for (int q = 0; q < sg.L; ++q) {
    cnt[q] = (int)picked_seg[b * sg.L + q];
    total += cnt[q];
    const int64_t* pq = pick_seg + ((size_t)b * sg.L + q) * pick_n_seg;
    for (int r = threadIdx.x; r < cnt[q]; r += 256) s_sc[q * pick_n_seg + r] = sb[pq[r]];
  }
  __syncthreads();
  const int64_t* pk = pick_seg + (size_t)blockIdx.x * pick_n_seg;
  for (int r = threadIdx.x; r < cnt[l]; r += 256) {
    const float sc = s_sc[l * pick_n_seg + r];
    int rank = r;
    for (int q = 0; q < sg.L; ++q) {
      if (q == l) continue;
      const float* a = s_sc + q * pick_n_seg;     // non-increasing
      int lo = 0, hi = cnt[q];
      if (q < l) {                                // entries with score >= sc come first
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] >= sc) lo = mid + 1; else hi = mid; }
      } else {                                    // entries with score > sc come first
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] > sc) lo = mid + 1; else hi = mid; }
      }
      rank += lo;
    }
    if (rank < top_k) pick[(size_t)b * top_k + rank] = pk[r];
  }
  if (l == 0) {
    // padding: the best candidate of the row = the best first survivor over the segments (ties: the lower segment); an empty row pads
    // with candidate 0 (what a stable sort of an all -inf key row puts first)
    int64_t best = 0;
    float bs = -INFINITY;
    bool any = false;
    for (int q = 0; q < sg.L; ++q)
      if (cnt[q] > 0 && (!any || s_sc[q * pick_n_seg] > bs)) {
        any = true;
        bs = s_sc[q * pick_n_seg];
        best = pick_seg[((size_t)b * sg.L + q) * pick_n_seg];
      }
    const int have = total < top_k ? total : top_k;
    for (int r = have + threadIdx.x; r < top_k; r += 256) pick[(size_t)b * top_k + r] = best;
    if (threadIdx.x == 0) picked[b] = have;
  }
}

__global__ void box_iou_kernel(const float* __restrict__ gt, int G, const float* __restrict__ boxes, int A, float* __restrict__ iou) {
  const int64_t total = (int64_t)G * A;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int a = (int)(i % A), g = (int)(i / A);
    iou[i] = box_iou_dev(gt + g * 4, boxes + (size_t)a * 4);
  }
}

// batched: iou[n][g][a] = IoU(gt[n][g], boxes[n][a]); boxes_stride = 0 shares one box set (anchors) between images
__global__ void box_iou_batched_kernel(const float* __restrict__ gt, int G, const float* __restrict__ boxes, int A, int N,
                                       long boxes_stride, float* __restrict__ iou) {
  const int64_t total = (int64_t)N * G * A;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int a = (int)(i % A);
    int64_t q = i / A;
    int g = (int)(q % G);
    int n = (int)(q / G);
    iou[i] = box_iou_dev(gt + ((size_t)n * G + g) * 4, boxes + (size_t)n * boxes_stride + (size_t)a * 4);
  }
}


// Radix-select step shared by the selection kernels: given the 256-bin histogram of the current digit, find the bin in which
// the `remaining`-th element (counting from the largest bin when DESC, from the smallest otherwise) falls, and how many of
// that bin's elements are still wanted.  Wave 0 does it with a lane-parallel prefix (4 bins per lane); the first version
// walked the bins in one thread -- 256 dependent LDS reads per pass, 10 us per pass, most of these kernels' time.
template <bool DESC>
__device__ __forceinline__ void radix_find_digit(const int* hist, int remaining, uint32_t prefix, int shift, uint32_t* s_prefix,
                                                 int* s_remaining) {
  if (threadIdx.x >= 64) return;
  const int l = threadIdx.x;
  int h[4], tot = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = hist[DESC ? 255 - (4 * l + j) : 4 * l + j];
    tot += h[j];
  }
  int incl = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (l >= d) incl += t;
  }
  const int excl = incl - tot;
  const bool hit = excl < remaining && remaining <= incl;
  const bool fallback = l == 63 && incl < remaining;          // cannot happen for consistent inputs; keeps the state defined
  if (hit || fallback) {
    int cum = excl, j = 0;
    for (; j < 3; ++j) {
      if (cum + h[j] >= remaining) break;
      cum += h[j];
    }
    const int bin = DESC ? 255 - (4 * l + j) : 4 * l + j;
    *s_prefix = prefix | ((uint32_t)bin << shift);
    *s_remaining = remaining - cum;
  }
}

// ---- per-row top-k by radix select + in-LDS sort -------------------------------------------------------------------------
// out = the indices of the k largest scores of a row segment in DESCENDING score order, equal scores by ASCENDING index:
// exactly what torch.sort(descending=True, stable=True)[1][:k] lists.  One 1024-thread block per (row, segment): 4 histogram
// passes of 8 bits over the segment (L2-resident) find the k-th largest key, one ordered pass collects the selected
// (key, index) pairs in LDS, a bitonic network orders them (k <= 4096).  Replaces torch.sort over [24, 16875] per level
// (rocprim merge sort: ~45 launches per step for the RPN levels).
__device__ __forceinline__ uint32_t order_key(float f) {
  uint32_t u = __float_as_uint(f == 0.f ? 0.f : f);        // -0.0 and +0.0 compare equal (ties go by index), as in torch.sort
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // larger float <=> larger key
}

struct TopkSegs {
  int nseg;
  int seg_off[8], n[8], k[8], out_off[8];
};

__global__ __launch_bounds__(1024) void topk_select_kernel(const float* __restrict__ scores, long row_stride, TopkSegs segs,
                                                           int64_t* __restrict__ out, long out_stride) {
  const int seg_off = segs.seg_off[blockIdx.y], n = segs.n[blockIdx.y], k = segs.k[blockIdx.y], out_off = segs.out_off[blockIdx.y];
  const int64_t idx_add = seg_off;
  __shared__ int hist[256];
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining;
  __shared__ int s_wave_g[16], s_wave_e[16];
  __shared__ uint64_t s_pair[4096];                          // (key << 32) | ~index: descending order = score desc, index asc
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = scores + (size_t)blockIdx.x * row_stride + seg_off;
  int64_t* orow = out + (size_t)blockIdx.x * out_stride + out_off;
  int P = 1;
  while (P < k) P <<= 1;
  if (k >= n) {                                              // everything is selected (k == n here: the host clamps k)
    for (int i = tid; i < n; i += 1024) s_pair[i] = ((uint64_t)order_key(row[i]) << 32) | (uint32_t)~i;
  } else {
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += 1024) {
        const uint32_t u = order_key(row[i]);
        if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
      }
      __syncthreads();
      radix_find_digit<true>(hist, remaining, prefix, shift, &s_prefix, &s_remaining);
      __syncthreads();
      prefix = s_prefix;
      remaining = s_remaining;
      mask |= 255u << shift;
      __syncthreads();
    }
    // prefix = key of the k-th largest element; `remaining` of the elements equal to it are taken, lowest indices first
    const uint32_t T = prefix;
    int base_g = 0, base_e = 0;                              // greater / equal elements in the chunks already done
    for (int i0 = 0; i0 < n; i0 += 1024) {
      const int i = i0 + tid;
      const uint32_t u = i < n ? order_key(row[i]) : 0u;
      const bool gt = i < n && u > T, eq = i < n && u == T;
      const uint64_t bg = __ballot(gt), be = __ballot(eq);
      const uint64_t below = (1ull << lane) - 1ull;
      if (lane == 0) {
        s_wave_g[wave] = __popcll(bg);
        s_wave_e[wave] = __popcll(be);
      }
      __syncthreads();
      int g_before = base_g + __popcll(bg & below), e_before = base_e + __popcll(be & below);
      int tot_g = 0, tot_e = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        if (w < wave) {
          g_before += s_wave_g[w];
          e_before += s_wave_e[w];
        }
        tot_g += s_wave_g[w];
        tot_e += s_wave_e[w];
      }
      if (gt || (eq && e_before < remaining)) s_pair[g_before + min(e_before, remaining)] = ((uint64_t)u << 32) | (uint32_t)~i;
      base_g += tot_g;
      base_e += tot_e;
      __syncthreads();
    }
  }
  for (int i = k + tid; i < P; i += 1024) s_pair[i] = 0;     // padding sorts to the end
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += 1024) {
        const int lo = 2 * t - (t & (stride - 1));           // index of the lower element of pair t
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;                  // direction of this bitonic block (final merge: descending)
        const uint64_t a = s_pair[lo], b = s_pair[hi];
        if ((a < b) == desc) {
          s_pair[lo] = b;
          s_pair[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += 1024) orow[i] = (int64_t)(uint32_t)~(uint32_t)s_pair[i] + idx_add;
}

// ---- fused box decoding (BoxCoder.decode_single + clip_boxes_to_image + validity tests) ----------------------------------
// The arithmetic follows the separate torch ops in order (fp-contract is off): dx = c*(1/wx), dw = min(c*(1/ww), clip) with NaN
// kept, pcx = dx*w + cx, pw = exp(dw)*w, corners = pc -/+ 0.5*pw, then clamp to [0, W] x [0, H].
// (torch divides a tensor by a python scalar as a multiplication by the fp32 reciprocal: the callers pass 1/weight)
__device__ __forceinline__ float4 decode_clip_dev(const float* c, const float* b, float iwx, float iwy, float iww, float iwh, float xclip,
                                                  float img_h, float img_w) {
  const float w = b[2] - b[0], h = b[3] - b[1];
  const float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
  const float dx = c[0] * iwx, dy = c[1] * iwy;
  float dw = c[2] * iww, dh = c[3] * iwh;
  dw = dw > xclip ? xclip : dw;
  dh = dh > xclip ? xclip : dh;
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  float4 o;
  o.x = pcx - 0.5f * pw;
  o.y = pcy - 0.5f * ph;
  o.z = pcx + 0.5f * pw;
  o.w = pcy + 0.5f * ph;
  // torch.clamp(min=0, max=L): NaN stays NaN
  o.x = o.x < 0.f ? 0.f : (o.x > img_w ? img_w : o.x);
  o.z = o.z < 0.f ? 0.f : (o.z > img_w ? img_w : o.z);
  o.y = o.y < 0.f ? 0.f : (o.y > img_h ? img_h : o.y);
  o.w = o.w < 0.f ? 0.f : (o.w > img_h ? img_h : o.w);
  return o;
}

// RPN: for image n and candidate t: a = top[n][t]; box = clip(decode(deltas[n][a], anchors[a])); prob = sigmoid(obj[n][a]);
// valid = w >= min_size && h >= min_size && prob >= score_thresh   (RegionProposalNetwork.filter_proposals [EXT])
__global__ void rpn_decode_filter_kernel(const float* __restrict__ deltas, const float* __restrict__ obj, const float* __restrict__ anchors,
                                         const int64_t* __restrict__ top, int N, int A, int K, float xclip, float img_h, float img_w,
                                         float min_size, float score_thresh, float* __restrict__ boxes, float* __restrict__ prob,
                                         uint8_t* __restrict__ valid) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N * K) return;
  const int n = j / K;
  const int64_t a = top[j];
  const float4 o = decode_clip_dev(deltas + ((size_t)n * A + a) * 4, anchors + (size_t)a * 4, 1.f, 1.f, 1.f, 1.f, xclip, img_h, img_w);
  const float x = obj[(size_t)n * A + a];
  const float p = 1.f / (1.f + expf(-x));
  *reinterpret_cast<float4*>(boxes + (size_t)j * 4) = o;
  prob[j] = p;
  valid[j] = ((o.z - o.x) >= min_size) && ((o.w - o.y) >= min_size) && (p >= score_thresh);
}

// RoI heads: codes [R][K*4], rois [R][4] -> boxes [R][K][4] clipped
__global__ void roi_decode_clip_kernel(const float* __restrict__ codes, const float* __restrict__ rois, long roi_stride, int R, int K,
                                       float wx, float wy, float ww, float wh, float xclip, float img_h, float img_w,
                                       float* __restrict__ boxes) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R * K) return;
  const int r = j / K;
  *reinterpret_cast<float4*>(boxes + (size_t)j * 4) = decode_clip_dev(codes + (size_t)j * 4, rois + (size_t)r * roi_stride, wx, wy, ww, wh, xclip,
                                                                      img_h, img_w);
}

// RoIHeads.postprocess_detections up to the NMS [EXT], for the fixed-size RoI list (row r belongs to image r / S and is real iff
// r % S < counts[image]): softmax over the C class logits, decode + clip of the K = C - 1 foreground boxes, and the candidate test
// (score > score_thresh, both sides >= min_size) in one launch.  Padding rows give zero boxes / scores and valid = 0.
__global__ void roi_postprocess_kernel(const float* __restrict__ logits, const float* __restrict__ codes, const float* __restrict__ rois,
                                       long roi_stride, const int64_t* __restrict__ counts, int R, int C, int S, float wx, float wy, float ww,
                                       float wh, float xclip, float img_h, float img_w, float score_thresh, float min_size,
                                       float* __restrict__ boxes, float* __restrict__ scores, uint8_t* __restrict__ valid) {
  const int K = C - 1;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R * K) return;
  const int r = j / K, k = j - r * K + 1;
  const bool real = (r % S) < (int)counts[r / S];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  float p = 0.f;
  bool v = false;
  if (real) {
    const float* lr = logits + (size_t)r * C;
    float mx = lr[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, lr[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lr[c] - mx);
    p = expf(lr[k] - mx) / se;
    o = decode_clip_dev(codes + ((size_t)r * C + k) * 4, rois + (size_t)r * roi_stride, wx, wy, ww, wh, xclip, img_h, img_w);
    v = (p > score_thresh) && ((o.z - o.x) >= min_size) && ((o.w - o.y) >= min_size);
  }
  *reinterpret_cast<float4*>(boxes + (size_t)j * 4) = o;
  scores[j] = p;
  valid[j] = v;
}

// ---- RoI sampling tail + FPN level mapping -------------------------------------------------------------------------------
// RoIHeads.select_training_samples after the sampler [EXT]: for the r-th selected candidate (flat index sel[r] into [N][T]):
// rois[r] = (image, box), labels[r], regression target = BoxCoder.encode(matched GT (zeros for GT-less images), box).
__global__ void roi_samples_finish_kernel(const int64_t* __restrict__ sel, int R, const float* __restrict__ comb, const int64_t* __restrict__ lab,
                                          const int64_t* __restrict__ matched, const float* __restrict__ gt, const uint8_t* __restrict__ gvalid,
                                          int T, int G, float wx, float wy, float ww, float wh, float* __restrict__ rois,
                                          int64_t* __restrict__ labels, float* __restrict__ reg_t) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int64_t s = sel[r];
  if (s < 0) {                     // padding row of the fixed-size RoI stage: sel = -(image + 1) -> empty box, label -1, zero target
    rois[(size_t)r * 5 + 0] = (float)(-s - 1);
    rois[(size_t)r * 5 + 1] = rois[(size_t)r * 5 + 2] = rois[(size_t)r * 5 + 3] = rois[(size_t)r * 5 + 4] = 0.f;
    labels[r] = -1;
    *reinterpret_cast<float4*>(reg_t + (size_t)r * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int n = (int)(s / T);
  const float4 b = *reinterpret_cast<const float4*>(comb + (size_t)s * 4);
  bool has_gt = false;
  for (int g = 0; g < G; ++g) has_gt = has_gt || gvalid[(size_t)n * G + g];
  const int64_t m = matched[s];
  float4 rg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_gt) rg = *reinterpret_cast<const float4*>(gt + ((size_t)n * G + (m > 0 ? m : 0)) * 4);
  rois[(size_t)r * 5 + 0] = (float)n;
  rois[(size_t)r * 5 + 1] = b.x;
  rois[(size_t)r * 5 + 2] = b.y;
  rois[(size_t)r * 5 + 3] = b.z;
  rois[(size_t)r * 5 + 4] = b.w;
  labels[r] = lab[s];
  const float ew = b.z - b.x, eh = b.w - b.y;
  const float ecx = b.x + 0.5f * ew, ecy = b.y + 0.5f * eh;
  const float gw = rg.z - rg.x, gh = rg.w - rg.y;
  const float gcx = rg.x + 0.5f * gw, gcy = rg.y + 0.5f * gh;
  float4 t;
  t.x = wx * (gcx - ecx) / ew;
  t.y = wy * (gcy - ecy) / eh;
  t.z = ww * logf(gw / ew);
  t.w = wh * logf(gh / eh);
  *reinterpret_cast<float4*>(reg_t + (size_t)r * 4) = t;
}

// Fixed-size form of the above, compaction included: one block per image scans its T candidates (selected = pos | neg), the r-th
// selected candidate (ascending candidate index: the order torchvision's torch.where gives) becomes row n*S + r, rows from the
// image's count up to S are padding (empty box, label -1, zero target); counts[n] = number of real rows.
__device__ __forceinline__ void roi_row_write(int r, int n, float4 b, int64_t label, float4 t, float* __restrict__ rois,
                                              int64_t* __restrict__ labels, float* __restrict__ reg_t) {
  rois[(size_t)r * 5 + 0] = (float)n;
  rois[(size_t)r * 5 + 1] = b.x;
  rois[(size_t)r * 5 + 2] = b.y;
  rois[(size_t)r * 5 + 3] = b.z;
  rois[(size_t)r * 5 + 4] = b.w;
  labels[r] = label;
  *reinterpret_cast<float4*>(reg_t + (size_t)r * 4) = t;
}

__global__ __launch_bounds__(256) void roi_samples_padded_kernel(const uint8_t* __restrict__ pos_sel, const uint8_t* __restrict__ neg_sel,
                                                                 const float* __restrict__ comb, const int64_t* __restrict__ lab,
                                                                 const int64_t* __restrict__ matched, const float* __restrict__ gt,
                                                                 const uint8_t* __restrict__ gvalid, int T, int G, int S, float wx, float wy,
                                                                 float ww, float wh, float* __restrict__ rois, int64_t* __restrict__ labels,
                                                                 float* __restrict__ reg_t, int64_t* __restrict__ counts) {
  __shared__ int wsum[4];
  __shared__ int carry;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  bool has_gt = false;
  for (int g = 0; g < G; ++g) has_gt = has_gt || gvalid[(size_t)n * G + g];
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    const size_t s = (size_t)n * T + t;
    const bool on = t < T && (pos_sel[s] | neg_sel[s]);
    const unsigned long long bal = __ballot(on);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int before = carry;
    for (int k = 0; k < w; ++k) before += wsum[k];
    const int r = before + in_wave;
    if (on && r < S) {
      const float4 b = *reinterpret_cast<const float4*>(comb + s * 4);
      const int64_t m = matched[s];
      float4 rg = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_gt) rg = *reinterpret_cast<const float4*>(gt + ((size_t)n * G + (m > 0 ? m : 0)) * 4);
      const float ew = b.z - b.x, eh = b.w - b.y;
      const float ecx = b.x + 0.5f * ew, ecy = b.y + 0.5f * eh;
      const float gw = rg.z - rg.x, gh = rg.w - rg.y;
      const float gcx = rg.x + 0.5f * gw, gcy = rg.y + 0.5f * gh;
      roi_row_write(n * S + r, n, b, lab[s], make_float4(wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * logf(gw / ew), wh * logf(gh / eh)), rois,
                    labels, reg_t);
    }
    __syncthreads();
    if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int cnt = carry < S ? carry : S;
  for (int r = cnt + tid; r < S; r += 256)
    roi_row_write(n * S + r, n, make_float4(0.f, 0.f, 0.f, 0.f), -1, make_float4(0.f, 0.f, 0.f, 0.f), rois, labels, reg_t);
  if (tid == 0) counts[n] = cnt;
}

// torchvision LevelMapper [EXT]: level = clamp(floor(k0 + log2(sqrt(area) / s0) + eps), k_min, k_max) - k_min
// (the division by the python scalar s0 is a multiplication by its fp32 reciprocal, as torch evaluates it)
__global__ void roi_levels_kernel(const float* __restrict__ rois, long stride, int R, float inv_s0, float k0, float eps, float k_min,
                                  float k_max, int* __restrict__ levels) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* b = rois + (size_t)r * stride;
  const float area = (b[2] - b[0]) * (b[3] - b[1]);
  float t = floorf((k0 + log2f(sqrtf(area) * inv_s0)) + eps);
  t = t < k_min ? k_min : (t > k_max ? k_max : t);
  levels[r] = (int)((long long)t - (long long)k_min);
}

// ---- BalancedPositiveNegativeSampler over N images in one launch ---------------------------------------------------------
// torchvision semantics [EXT]: per image take min(P, cap_p) random positives and min(Nneg, B - num_pos) random negatives.
// A uniformly random subset of size k = the k smallest of iid random keys: `keys` holds one random int32 >= 0 per
// candidate (torch.randint, so the generator stream is the framework's); a block radix-selects the k-th smallest key of
// each class and marks the members (equal keys at the cut: lowest index first).
// `lds_keys` / `lds_cls` (optional): the row's keys and classes (1 positive, 0 negative, 2 neither) staged in LDS by the caller -- the
// four radix passes and the marking pass then read LDS instead of streaming 12 bytes per anchor from L2 five times.
__device__ __forceinline__ void select_smallest_marked(const int32_t* __restrict__ keys, const int64_t* __restrict__ labels, int A, bool want_pos,
                                                       int k, int population, uint8_t* __restrict__ out, int* hist, uint32_t* s_prefix,
                                                       int* s_remaining, int* s_wave_l, int* s_wave_e, const uint32_t* lds_keys = nullptr,
                                                       const uint8_t* lds_cls = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto member = [&](int i) {
    if (lds_cls) return lds_cls[i] == (want_pos ? 1 : 0);
    const int64_t l = labels[i];
    return want_pos ? (l >= 1) : (l == 0);
  };
  auto key_of = [&](int i) { return lds_keys ? lds_keys[i] : (uint32_t)keys[i]; };
  if (k <= 0) {
    for (int i = tid; i < A; i += 1024) out[i] = 0;
    return;
  }
  if (k >= population) {
    for (int i = tid; i < A; i += 1024) out[i] = member(i) ? 1 : 0;
    return;
  }
  uint32_t prefix = 0, mask = 0;
  int remaining = k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < A; i += 1024) {
      if (!member(i)) continue;
      const uint32_t u = key_of(i);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
    }
    __syncthreads();
    radix_find_digit<false>(hist, remaining, prefix, shift, s_prefix, s_remaining);
    __syncthreads();
    prefix = *s_prefix;
    remaining = *s_remaining;
    mask |= 255u << shift;
    __syncthreads();
  }
  const uint32_t T = prefix;                  // the k-th smallest key; `remaining` of the members equal to it are taken
  int base_e = 0;
  for (int i0 = 0; i0 < A; i0 += 1024) {
    const int i = i0 + tid;
    const bool mem = i < A && member(i);
    const uint32_t u = mem ? key_of(i) : 0xFFFFFFFFu;
    const bool lt = mem && u < T, eq = mem && u == T;
    const uint64_t be = __ballot(eq);
    if (lane == 0) s_wave_e[wave] = __popcll(be);
    __syncthreads();
    int e_before = base_e + __popcll(be & ((1ull << lane) - 1ull));
    int tot_e = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) e_before += s_wave_e[w];
      tot_e += s_wave_e[w];
    }
    if (i < A) out[i] = (lt || (eq && e_before < remaining)) ? 1 : 0;
    base_e += tot_e;
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void sample_pos_neg_kernel(const int64_t* __restrict__ labels, const int32_t* __restrict__ keys, int A,
                                                              int batch_size, int cap_pos, uint8_t* __restrict__ pos_sel,
                                                              uint8_t* __restrict__ neg_sel, int64_t* __restrict__ counts, int use_lds) {
  __shared__ int hist[256];
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining;
  __shared__ int s_wave_l[16], s_wave_e[16];
  __shared__ int s_P, s_N;
  extern __shared__ uint32_t s_dyn[];          // [A] keys, then [A] class bytes -- when the launch was given the room (use_lds)
  const int n = blockIdx.x, tid = threadIdx.x;
  const int64_t* lb = labels + (size_t)n * A;
  const int32_t* kb = keys + (size_t)n * A;
  uint32_t* lk = use_lds ? s_dyn : nullptr;
  uint8_t* lc = use_lds ? reinterpret_cast<uint8_t*>(s_dyn + A) : nullptr;
  if (tid == 0) s_P = s_N = 0;
  __syncthreads();
  int p = 0, q = 0;
  for (int i = tid; i < A; i += 1024) {
    const int64_t l = lb[i];
    p += l >= 1;
    q += l == 0;
    if (use_lds) {
      lk[i] = (uint32_t)kb[i];
      lc[i] = l >= 1 ? 1 : (l == 0 ? 0 : 2);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    p += __shfl_xor(p, d);
    q += __shfl_xor(q, d);
  }
  if ((tid & 63) == 0) {
    atomicAdd(&s_P, p);
    atomicAdd(&s_N, q);
  }
  __syncthreads();
  const int P = s_P, Nn = s_N;
  const int num_pos = min(P, cap_pos);
  const int num_neg = min(Nn, batch_size - num_pos);
  if (tid == 0) {
    counts[(size_t)n * 2 + 0] = num_pos;
    counts[(size_t)n * 2 + 1] = num_neg;
  }
  select_smallest_marked(kb, lb, A, true, num_pos, P, pos_sel + (size_t)n * A, hist, &s_prefix, &s_remaining, s_wave_l, s_wave_e, lk, lc);
  __syncthreads();
  select_smallest_marked(kb, lb, A, false, num_neg, Nn, neg_sel + (size_t)n * A, hist, &s_prefix, &s_remaining, s_wave_l, s_wave_e, lk, lc);
}

// ---- fused target assignment (box_iou + Matcher + label lookup + BoxCoder.encode), one thread per (image, box) ---------
// Replaces ~70 elementwise launches over [N, G, A] / [N, A] tensors per call (detection.py: _match_batched and its callers).
// Arithmetic order follows the separate torch ops (fp-contract is off for this library), so the results are the ones
// the per-op path produces: IoU via box_iou_dev, first maximal GT on ties (torch.max), torchvision's low-quality rule
// (an anchor whose IoU equals some GT's best IoU gets ITS OWN arg-max GT, not that GT), encode = BoxCoder.encode_single.

// pass 1 (allow_low_quality only): best[n][g] = max over boxes of IoU(gt[n][g], box); -1 for invalid GT rows
// (1 024 threads: a block per (image, GT) is all the parallelism there is -- 192 blocks for the RPN's 22 743 anchors -- so the chain of
// dependent anchor loads per thread sets the time: 89 trips with 256 threads, 23 with 1 024.  max is order-independent: same result.)
__global__ __launch_bounds__(1024) void match_best_kernel(const float* __restrict__ gt, const uint8_t* __restrict__ gvalid, int G,
                                                          const float* __restrict__ boxes, int A, long boxes_stride,
                                                          float* __restrict__ best) {
  const int ng = blockIdx.x;                 // n * G + g
  const int n = ng / G;
  __shared__ float red[16];
  float m = -1.f;
  if (gvalid[ng]) {
    const float g4[4] = {gt[(size_t)ng * 4 + 0], gt[(size_t)ng * 4 + 1], gt[(size_t)ng * 4 + 2], gt[(size_t)ng * 4 + 3]};
    const float* bb = boxes + (size_t)n * boxes_stride;
    for (int a = threadIdx.x; a < A; a += 1024) m = fmaxf(m, box_iou_dev(g4, bb + (size_t)a * 4));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) r = fmaxf(r, red[w]);
    best[ng] = r;
  }
}

__global__ __launch_bounds__(256) void match_assign_kernel(const float* __restrict__ gt, const uint8_t* __restrict__ gvalid,
                                                           const int64_t* __restrict__ glabels, int G, const float* __restrict__ boxes,
                                                           int A, int N, long boxes_stride, float high, float low,
                                                           const float* __restrict__ best, float wx, float wy, float ww, float wh,
                                                           int want_reg, int64_t* __restrict__ matched, int64_t* __restrict__ labels,
                                                           float* __restrict__ reg_t) {
  extern __shared__ float s_gt[];            // [G][4] boxes, [G] best, [G] valid (as float)
  const int n = blockIdx.y;
  float* s_best = s_gt + G * 4;
  float* s_valid = s_best + G;
  for (int i = threadIdx.x; i < G * 4; i += 256) s_gt[i] = gt[(size_t)n * G * 4 + i];
  for (int i = threadIdx.x; i < G; i += 256) {
    s_best[i] = best ? best[n * G + i] : 0.f;
    s_valid[i] = gvalid[n * G + i] ? 1.f : 0.f;
  }
  __syncthreads();
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= A) return;
  const float* bp = boxes + (size_t)n * boxes_stride + (size_t)a * 4;
  const float b4[4] = {bp[0], bp[1], bp[2], bp[3]};
  float vmax = -INFINITY;
  int imax = 0;
  bool is_best = false, has_gt = false;
  for (int g = 0; g < G; ++g) {
    const bool v = s_valid[g] != 0.f;
    has_gt = has_gt || v;
    const float iou = v ? box_iou_dev(s_gt + g * 4, b4) : -1.f;
    if (iou > vmax) {                        // strict: the FIRST maximal GT wins, as torch.max(dim) does
      vmax = iou;
      imax = g;
    }
    if (best && v && iou == s_best[g]) is_best = true;
  }
  int64_t m = imax;
  if (vmax < low) m = -1;                                     // Matcher.BELOW_LOW_THRESHOLD
  else if (vmax < high) m = -2;                               // Matcher.BETWEEN_THRESHOLDS
  if (is_best) m = imax;
  const size_t o = (size_t)n * A + a;
  matched[o] = m;
  if (labels) {
    int64_t lab = m >= 0 ? (glabels ? glabels[(size_t)n * G + m] : 1) : (m == -1 ? 0 : -1);
    labels[o] = has_gt ? lab : 0;
  }
  if (want_reg) {
    const float* r = s_gt + (m > 0 ? (int)m : 0) * 4;         // matched.clamp(min=0)
    const float ew = b4[2] - b4[0], eh = b4[3] - b4[1];
    const float ecx = b4[0] + 0.5f * ew, ecy = b4[1] + 0.5f * eh;
    const float gw = r[2] - r[0], gh = r[3] - r[1];
    const float gcx = r[0] + 0.5f * gw, gcy = r[1] + 0.5f * gh;
    float4 t;
    t.x = wx * (gcx - ecx) / ew;
    t.y = wy * (gcy - ecy) / eh;
    t.z = ww * logf(gw / ew);
    t.w = wh * logf(gh / eh);
    *reinterpret_cast<float4*>(reg_t + o * 4) = t;
  }
}

}  // namespace

extern "C" int hd_box_iou_batched(const float* gt, int G, const float* boxes, int A, int N, int shared_boxes, float* iou, void* stream) {
  HD_CHECK_ARG(gt && boxes && iou && G >= 0 && A >= 0 && N >= 0, "hd_box_iou_batched: bad args");
  if (G == 0 || A == 0 || N == 0) return HD_OK;
  int64_t total = (int64_t)N * G * A;
  int g = (int)((total + 255) / 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(box_iou_batched_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, gt, G, boxes, A, N,
                     shared_boxes ? 0l : (long)A * 4, iou);
  HD_CHECK_LAUNCH();
  return HD_OK;
}


extern "C" int hd_rpn_decode_filter(const float* deltas, const float* objectness, const float* anchors, const int64_t* top, int N, int A, int K,
                                    float bbox_xform_clip, float img_h, float img_w, float min_size, float score_thresh, float* boxes,
                                    float* prob, uint8_t* valid, void* stream) {
  HD_CHECK_ARG(deltas && objectness && anchors && top && boxes && prob && valid && N >= 0 && A >= 0 && K >= 0, "hd_rpn_decode_filter: bad args");
  if (N * K == 0) return HD_OK;
  hipLaunchKernelGGL(rpn_decode_filter_kernel, dim3((N * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, deltas, objectness, anchors, top, N,
                     A, K, bbox_xform_clip, img_h, img_w, min_size, score_thresh, boxes, prob, valid);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_roi_decode_clip(const float* codes, const float* rois, long roi_stride, int R, int K, const float* coder_weights,
                                  float bbox_xform_clip, float img_h, float img_w, float* boxes, void* stream) {
  HD_CHECK_ARG(codes && rois && coder_weights && boxes && R >= 0 && K >= 0 && roi_stride >= 4, "hd_roi_decode_clip: bad args");
  if (R * K == 0) return HD_OK;
  hipLaunchKernelGGL(roi_decode_clip_kernel, dim3((R * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, codes, rois, roi_stride, R, K,
                     1.0f / coder_weights[0], 1.0f / coder_weights[1], 1.0f / coder_weights[2], 1.0f / coder_weights[3], bbox_xform_clip, img_h,
                     img_w, boxes);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_roi_postprocess(const float* class_logits, const float* box_regression, const float* rois, long roi_stride, const int64_t* counts,
                                  int R, int C, int S, const float* coder_weights, float bbox_xform_clip, float img_h, float img_w, float score_thresh,
                                  float min_size, float* boxes, float* scores, uint8_t* valid, void* stream) {
  HD_CHECK_ARG(class_logits && box_regression && rois && counts && coder_weights && boxes && scores && valid && R >= 0 && C >= 2 && S > 0 &&
               roi_stride >= 4, "hd_roi_postprocess: bad args");
  if (R == 0) return HD_OK;
  const int total = R * (C - 1);
  hipLaunchKernelGGL(roi_postprocess_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, class_logits, box_regression, rois,
                     roi_stride, counts, R, C, S, 1.0f / coder_weights[0], 1.0f / coder_weights[1], 1.0f / coder_weights[2], 1.0f / coder_weights[3],
                     bbox_xform_clip, img_h, img_w, score_thresh, min_size, boxes, scores, valid);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_roi_samples_finish(const int64_t* sel, int R, const float* comb, const int64_t* lab, const int64_t* matched, const float* gt,
                                     const uint8_t* gvalid, int T, int G, const float* coder_weights, float* rois, int64_t* labels, float* reg_t,
                                     void* stream) {
  HD_CHECK_ARG(R >= 0 && T > 0 && G > 0 && coder_weights, "hd_roi_samples_finish: bad args");
  if (R == 0) return HD_OK;
  HD_CHECK_ARG(sel && comb && lab && matched && gt && gvalid && rois && labels && reg_t, "hd_roi_samples_finish: null pointer");
  hipLaunchKernelGGL(roi_samples_finish_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, sel, R, comb, lab, matched, gt, gvalid, T,
                     G, coder_weights[0], coder_weights[1], coder_weights[2], coder_weights[3], rois, labels, reg_t);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_roi_samples_padded(const uint8_t* pos_sel, const uint8_t* neg_sel, const float* comb, const int64_t* lab, const int64_t* matched,
                                     const float* gt, const uint8_t* gvalid, int N, int T, int G, int S, const float* coder_weights, float* rois,
                                     int64_t* labels, float* reg_t, int64_t* counts, void* stream) {
  HD_CHECK_ARG(pos_sel && neg_sel && comb && lab && matched && gt && gvalid && coder_weights && rois && labels && reg_t && counts && N > 0 &&
               T > 0 && G > 0 && S > 0, "hd_roi_samples_padded: bad args");
  hipLaunchKernelGGL(roi_samples_padded_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, pos_sel, neg_sel, comb, lab, matched, gt, gvalid, T, G, S,
                     coder_weights[0], coder_weights[1], coder_weights[2], coder_weights[3], rois, labels, reg_t, counts);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_roi_levels(const float* boxes, long stride, int R, float canonical_scale, float canonical_level, float eps, int k_min, int k_max,
                             int* levels, void* stream) {
  HD_CHECK_ARG(R >= 0 && stride >= 4 && k_max >= k_min && canonical_scale > 0.f, "hd_roi_levels: bad args");
  if (R == 0) return HD_OK;
  HD_CHECK_ARG(boxes && levels, "hd_roi_levels: null pointer");
  hipLaunchKernelGGL(roi_levels_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, stride, R, 1.0f / canonical_scale,
                     canonical_level, eps, (float)k_min, (float)k_max, levels);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_sample_pos_neg(const int64_t* labels, const int32_t* keys, int N, int A, int batch_size, int cap_pos, uint8_t* pos_sel,
                                 uint8_t* neg_sel, int64_t* counts, void* stream) {
  HD_CHECK_ARG(N >= 0 && A >= 0 && batch_size > 0 && cap_pos >= 0 && cap_pos <= batch_size, "hd_sample_pos_neg: bad args");
  if (N == 0 || A == 0) return HD_OK;
  HD_CHECK_ARG(labels && keys && pos_sel && neg_sel && counts, "hd_sample_pos_neg: null pointer");
  // keys + classes of a row in LDS when they fit (5 bytes per anchor; the RPN's 22 743 anchors: 111 KB): the radix passes read LDS
  const size_t need = (size_t)A * 4 + (((size_t)A + 3) & ~(size_t)3);
  const int use_lds = need <= 150 * 1024 ? 1 : 0;
  hipLaunchKernelGGL(sample_pos_neg_kernel, dim3(N), dim3(1024), use_lds ? need : 0, (hipStream_t)stream, labels, keys, A, batch_size, cap_pos,
                     pos_sel, neg_sel, counts, use_lds);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_topk_select_rows(const float* scores, int B, long row_stride, const int* seg_sizes, int nseg, int k, int64_t* out,
                                   long out_stride, void* stream) {
  HD_CHECK_ARG(scores && out && seg_sizes && B >= 0 && nseg >= 1 && nseg <= 8 && k >= 1, "hd_topk_select_rows: bad args (1 <= nseg <= 8)");
  TopkSegs sg;
  sg.nseg = nseg;
  int off = 0, ooff = 0;
  for (int i = 0; i < nseg; ++i) {
    const int n = seg_sizes[i], kk = k < n ? k : n;
    HD_CHECK_ARG(n >= 1 && kk <= 4096, "hd_topk_select_rows: segments must be non-empty, at most 4096 selected entries each (k=%d, n=%d)", k, n);
    sg.seg_off[i] = off; sg.n[i] = n; sg.k[i] = kk; sg.out_off[i] = ooff;
    off += n;
    ooff += kk;
  }
  if (B == 0) return HD_OK;
  hipLaunchKernelGGL(topk_select_kernel, dim3(B, nseg), dim3(1024), 0, (hipStream_t)stream, scores, row_stride, sg, out, out_stride);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_match_targets(const float* gt, const uint8_t* gvalid, const int64_t* glabels, int G, const float* boxes, int A, int N,
                                int shared_boxes, float high, float low, int allow_low_quality, const float* coder_weights,
                                float* best_ws, int64_t* matched, int64_t* labels, float* reg_t, void* stream) {
  HD_CHECK_ARG(gt && gvalid && boxes && matched && G >= 1 && G <= 2048 && A >= 0 && N >= 0, "hd_match_targets: bad args (1 <= G <= 2048)");
  HD_CHECK_ARG(!allow_low_quality || best_ws, "hd_match_targets: allow_low_quality needs the [N*G] float workspace");
  HD_CHECK_ARG((reg_t == nullptr) == (coder_weights == nullptr), "hd_match_targets: reg_t and coder_weights go together");
  if (A == 0 || N == 0) return HD_OK;
  hipStream_t s = (hipStream_t)stream;
  const long stride = shared_boxes ? 0l : (long)A * 4;
  if (allow_low_quality)
    hipLaunchKernelGGL(match_best_kernel, dim3(N * G), dim3(1024), 0, s, gt, gvalid, G, boxes, A, stride, best_ws);
  const float* w = coder_weights;
  hipLaunchKernelGGL(match_assign_kernel, dim3((A + 255) / 256, N), dim3(256), (size_t)G * 6 * sizeof(float), s, gt, gvalid, glabels, G, boxes,
                     A, N, stride, high, low, allow_low_quality ? (const float*)best_ws : (const float*)nullptr, w ? w[0] : 1.f, w ? w[1] : 1.f,
                     w ? w[2] : 1.f, w ? w[3] : 1.f, reg_t ? 1 : 0, matched, labels, reg_t);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_nms_sorted_batched(const float* boxes, const int* counts, int B, int nmax, float iou_thr, uint64_t* mask_ws,
                                     uint8_t* keep, void* stream) {
  return hd_nms_sorted_batched_topk(boxes, counts, B, nmax, iou_thr, mask_ws, keep, 0x7fffffff, stream);
}

extern "C" int hd_nms_sorted_batched_topk(const float* boxes, const int* counts, int B, int nmax, float iou_thr, uint64_t* mask_ws,
                                          uint8_t* keep, int max_keep, void* stream) {
  HD_CHECK_ARG(boxes && counts && mask_ws && keep && B > 0 && nmax > 0 && nmax <= 16384 && max_keep > 0, "hd_nms_sorted_batched: bad args (nmax<=16384)");
  hipStream_t s = (hipStream_t)stream;
  int cb = (nmax + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, B), dim3(64), 0, s, boxes, counts, nmax, iou_thr, mask_ws);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(256), 0, s, (const uint64_t*)mask_ws, counts, nmax, keep, max_keep,
                     (const int64_t*)nullptr, (int64_t*)nullptr, 0, (int64_t*)nullptr);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_batched_nms_pick(const float* boxes, const int64_t* idxs, const uint8_t* valid, const int64_t* order, int B, int n,
                                   float iou_thr, int top_n, float* sorted_ws, int* counts_ws, uint64_t* mask_ws, uint8_t* keep_ws,
                                   int64_t* pick, int64_t* picked, void* stream) {
  HD_CHECK_ARG(boxes && idxs && valid && order && sorted_ws && counts_ws && mask_ws && keep_ws && pick && picked && B > 0 && n > 0 &&
               n <= 16384 && top_n > 0, "hd_batched_nms_pick: bad args (n<=16384)");
  hipStream_t s = (hipStream_t)stream;
  const int cb = (n + 63) / 64;
  hipLaunchKernelGGL(nms_prepare_kernel, dim3(B), dim3(256), 0, s, boxes, idxs, valid, order, n, sorted_ws, counts_ws);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, B), dim3(64), 0, s, (const float*)sorted_ws, (const int*)counts_ws, n, iou_thr, mask_ws);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(256), 0, s, (const uint64_t*)mask_ws, (const int*)counts_ws, n, keep_ws, top_n, order, pick,
                     top_n < n ? top_n : n, picked);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_batched_nms_pick_segments(const float* boxes, const float* scores, const uint8_t* valid, int B, int n, const int* seg_sizes,
                                            int L, float iou_thr, int top_n, float* sorted_ws, int64_t* order_ws, int* counts_ws,
                                            uint64_t* mask_ws, uint8_t* keep_ws, int64_t* pick_ws, int64_t* picked_seg, int64_t* pick,
                                            int64_t* picked, void* stream) {
  HD_CHECK_ARG(boxes && scores && valid && seg_sizes && sorted_ws && order_ws && counts_ws && mask_ws && keep_ws && pick_ws && picked_seg && pick && picked &&
               B > 0 && n > 0 && L >= 1 && L <= 8 && top_n > 0, "hd_batched_nms_pick_segments: bad args (<= 8 segments)");
  NmsSegs sg;
  sg.L = L;
  sg.S = 1;
  sg.off[0] = 0;
  for (int l = 0; l < L; ++l) {
    HD_CHECK_ARG(seg_sizes[l] > 0, "hd_batched_nms_pick_segments: empty segment");
    sg.off[l + 1] = sg.off[l] + seg_sizes[l];
    if (seg_sizes[l] > sg.S) sg.S = seg_sizes[l];
  }
  for (int l = L + 1; l < 9; ++l) sg.off[l] = sg.off[L];
  HD_CHECK_ARG(sg.off[L] == n && sg.S <= 16384, "hd_batched_nms_pick_segments: segment sizes must add up to n");
  hipStream_t s = (hipStream_t)stream;
  const int S = sg.S, cb = (S + 63) / 64, pick_n = top_n < S ? top_n : S;
  hipLaunchKernelGGL(nms_seg_prepare_kernel, dim3(B), dim3(256), 0, s, boxes, valid, n, sg, sorted_ws, order_ws, counts_ws);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, B * L), dim3(64), 0, s, (const float*)sorted_ws, (const int*)counts_ws, S, iou_thr, mask_ws);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(B * L), dim3(256), 0, s, (const uint64_t*)mask_ws, (const int*)counts_ws, S, keep_ws, top_n,
                     (const int64_t*)order_ws, pick_ws, pick_n, picked_seg);
  const int top_k = top_n < n ? top_n : n;
  hipLaunchKernelGGL(nms_seg_merge_kernel, dim3(B * L), dim3(256), (size_t)L * pick_n * sizeof(float), s, scores, (const int64_t*)pick_ws,
                     (const int64_t*)picked_seg, n, pick_n, sg, top_k, pick, picked);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_box_iou(const float* gt, int G, const float* boxes, int A, float* iou, void* stream) {
  HD_CHECK_ARG(gt && boxes && iou && G >= 0 && A >= 0, "hd_box_iou: bad args");
  if (G == 0 || A == 0) return HD_OK;
  int64_t total = (int64_t)G * A;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(box_iou_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, gt, G, boxes, A, iou);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
