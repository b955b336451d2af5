// dev microbenchmark / feasibility study for the next round ("what is left"):
// the 64x64 MLP chain of the node update  X <- relu(W_l X + b_l)  (64 features x 32 nodes per wave-tile, weights in LDS,
// activations in accumulator registers, the accumulators of one layer being the B operand of the next) evaluated
//   (a) on v_mfma_f32_32x32x2_f32 (exact fp32; what the scorer does today, 64 MFMAs of 64 cycles per layer), and
//   (b) on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into three bf16 pieces, x = x1 + x2 + x3 (24 mantissa
//       bits), and the six products of total order <= 4 summed in the fp32 accumulator: w1x1 + w1x2 + w2x1 + w1x3 + w2x2 +
//       w3x1 -- 48 MFMAs of 32 cycles per layer (a quarter of the matrix-pipe time), plus the VALU work of splitting the
//       activations of every layer (the weights are split once on the host).
// Prints the time per layer-tile of both forms on the whole chip and their error against an fp64 evaluation.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int LAYERS = 4;          // layers resident in LDS (re-used cyclically by the timing loop)
constexpr int WAVES = 12;          // waves per workgroup, as k_node_update

// feature held by accumulator tile `it`, register `reg`, lane half `h`  (C/D layout of every 32x32 MFMA)
__host__ __device__ inline int feat_of(int it, int reg, int h) { return 32 * it + (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- (a) fp32 MFMA: k-step s of the B operand = accumulator register s of tile s / 16 (lane half h holds feature
// feat_of(s / 16, s % 16, h)); A operand: lane (r, h) = W[32 ot + r][that feature]
__global__ __launch_bounds__(WAVES * 64) void k_chain_f32(const float* wpack, const float* bias, const float* x_in, float* x_out, int tiles_per_wave,
                                                          int layers_per_tile) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < LAYERS * 2 * 32 * 64; i += blockDim.x) lds[i] = wpack[i];
  float* lb = lds + LAYERS * 2 * 32 * 64;
  for (int i = threadIdx.x; i < LAYERS * 64; i += blockDim.x) lb[i] = bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * WAVES + wave;
  for (int t = 0; t < tiles_per_wave; ++t) {
    const long tile = wid * tiles_per_wave + t;
    f32x16 X[2];
    for (int it = 0; it < 2; ++it)
      for (int q = 0; q < 16; ++q) X[it][q] = x_in[(tile * 32 + r) * 64 + feat_of(it, q, h)];
    for (int l = 0; l < layers_per_tile; ++l) {
      const int ll = l % LAYERS;
      const float* w = lds + ll * 2 * 32 * 64;
      f32x16 Y[2];
      for (int ot = 0; ot < 2; ++ot)
        for (int q = 0; q < 16; ++q) Y[ot][q] = lb[ll * 64 + feat_of(ot, q, h)];
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const float b = X[s >> 4][s & 15];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) Y[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[(ot * 32 + s) * 64 + lane], b, Y[ot], 0, 0, 0);
      }
      for (int ot = 0; ot < 2; ++ot)
        for (int q = 0; q < 16; ++q) X[ot][q] = fmaxf(Y[ot][q], 0.0f);
    }
    for (int it = 0; it < 2; ++it)
      for (int q = 0; q < 16; ++q) x_out[(tile * 32 + r) * 64 + feat_of(it, q, h)] = X[it][q];
  }
}

// ---- (b) bf16 x 3
__device__ __forceinline__ unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ float lo_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

struct Split { unsigned p[3][4]; };          // three bf16x8 fragments (k-step of 8 features per lane half)
// registers q0 .. q0+7 of an accumulator tile -> three bf16 pieces of each
__device__ __forceinline__ Split split8(const f32x16& t, int q0) {
  Split s;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = t[q0 + 2 * j], b = t[q0 + 2 * j + 1];
    const unsigned p1 = pk_bf16(a, b);
    const float ra = a - lo_f32(p1), rb = b - hi_f32(p1);
    const unsigned p2 = pk_bf16(ra, rb);
    const float sa = ra - lo_f32(p2), sb = rb - hi_f32(p2);
    s.p[0][j] = p1; s.p[1][j] = p2; s.p[2][j] = pk_bf16(sa, sb);
  }
  return s;
}
__device__ __forceinline__ bf16x8 frag(const unsigned (&p)[4]) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(bf16x8, u32x4{p[0], p[1], p[2], p[3]});
}

// weights: [layer][piece 3][ot 2][kstep 4][lane 64] x 16 bytes
__global__ __launch_bounds__(WAVES * 64) void k_chain_bf3(const uint4* wpack, const float* bias, const float* x_in, float* x_out, int tiles_per_wave,
                                                          int layers_per_tile) {
  extern __shared__ float lds[];
  uint4* lw = reinterpret_cast<uint4*>(lds);
  for (int i = threadIdx.x; i < LAYERS * 3 * 2 * 4 * 64; i += blockDim.x) lw[i] = wpack[i];
  float* lb = lds + LAYERS * 3 * 2 * 4 * 64 * 4;
  for (int i = threadIdx.x; i < LAYERS * 64; i += blockDim.x) lb[i] = bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * WAVES + wave;
  for (int t = 0; t < tiles_per_wave; ++t) {
    const long tile = wid * tiles_per_wave + t;
    f32x16 X[2];
    for (int it = 0; it < 2; ++it)
      for (int q = 0; q < 16; ++q) X[it][q] = x_in[(tile * 32 + r) * 64 + feat_of(it, q, h)];
    for (int l = 0; l < layers_per_tile; ++l) {
      const int ll = l % LAYERS;
      const uint4* w = lw + ll * 3 * 2 * 4 * 64;
      f32x16 Y[2];
      for (int ot = 0; ot < 2; ++ot)
        for (int q = 0; q < 16; ++q) Y[ot][q] = lb[ll * 64 + feat_of(ot, q, h)];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                 // k-step ks: registers 8 (ks & 1) .. + 7 of tile ks >> 1
        const Split x = split8(X[ks >> 1], 8 * (ks & 1));
        const bf16x8 x1 = frag(x.p[0]), x2 = frag(x.p[1]), x3 = frag(x.p[2]);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          const bf16x8 w1 = __builtin_bit_cast(bf16x8, w[((0 * 2 + ot) * 4 + ks) * 64 + lane]);
          const bf16x8 w2 = __builtin_bit_cast(bf16x8, w[((1 * 2 + ot) * 4 + ks) * 64 + lane]);
          const bf16x8 w3 = __builtin_bit_cast(bf16x8, w[((2 * 2 + ot) * 4 + ks) * 64 + lane]);
          // smallest terms first
          Y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x1, Y[ot], 0, 0, 0);
          Y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x2, Y[ot], 0, 0, 0);
          Y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x3, Y[ot], 0, 0, 0);
          Y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x1, Y[ot], 0, 0, 0);
          Y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x2, Y[ot], 0, 0, 0);
          Y[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, Y[ot], 0, 0, 0);
        }
      }
      for (int ot = 0; ot < 2; ++ot)
        for (int q = 0; q < 16; ++q) X[ot][q] = fmaxf(Y[ot][q], 0.0f);
    }
    for (int it = 0; it < 2; ++it)
      for (int q = 0; q < 16; ++q) x_out[(tile * 32 + r) * 64 + feat_of(it, q, h)] = X[it][q];
  }
}

// ---- host
static unsigned short bf16_rne(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f32(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  const int n_cu = 256, tiles_per_wave = 3;
  const long waves = (long)n_cu * WAVES, tiles = waves * tiles_per_wave, nodes = tiles * 32;
  std::vector<float> W(LAYERS * 64 * 64), B(LAYERS * 64), X(nodes * 64);
  srand(7);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  for (auto& v : W) v = rnd() * 0.25f;      // ~ N(0, 1/fan_in) scale
  for (auto& v : B) v = rnd() * 0.1f;
  for (auto& v : X) v = rnd();
  // (a) pack: [layer][ot][s 32][lane 64]
  std::vector<float> wa(LAYERS * 2 * 32 * 64);
  for (int l = 0; l < LAYERS; ++l)
    for (int ot = 0; ot < 2; ++ot)
      for (int s = 0; s < 32; ++s)
        for (int lane = 0; lane < 64; ++lane)
          wa[((l * 2 + ot) * 32 + s) * 64 + lane] = W[(l * 64 + 32 * ot + (lane & 31)) * 64 + feat_of(s >> 4, s & 15, lane >> 5)];
  // (b) pack: [layer][piece][ot][ks][lane][8 bf16]: element j = W[32 ot + r][feat_of(ks >> 1, 8 (ks & 1) + j, h)]
  std::vector<unsigned short> wb((size_t)LAYERS * 3 * 2 * 4 * 64 * 8);
  for (int l = 0; l < LAYERS; ++l)
    for (int ot = 0; ot < 2; ++ot)
      for (int ks = 0; ks < 4; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            float w = W[(l * 64 + 32 * ot + (lane & 31)) * 64 + feat_of(ks >> 1, 8 * (ks & 1) + j, lane >> 5)];
            for (int p = 0; p < 3; ++p) {
              const unsigned short b = bf16_rne(w);
              wb[(((((size_t)l * 3 + p) * 2 + ot) * 4 + ks) * 64 + lane) * 8 + j] = b;
              w -= bf16_f32(b);
            }
          }
  float *dwa, *db, *dx, *dy;
  uint4* dwb;
  hipMalloc(&dwa, wa.size() * 4); hipMalloc(&db, B.size() * 4); hipMalloc(&dx, X.size() * 4); hipMalloc(&dy, X.size() * 4);
  hipMalloc(&dwb, wb.size() * 2);
  hipMemcpy(dwa, wa.data(), wa.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  const size_t lds_a = (LAYERS * 2 * 32 * 64 + LAYERS * 64) * 4, lds_b = (size_t)LAYERS * 3 * 2 * 4 * 64 * 16 + LAYERS * 64 * 4;
  hipFuncSetAttribute((const void*)k_chain_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
  hipFuncSetAttribute((const void*)k_chain_bf3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);

  // ---- numerics: LAYERS layers on the first 64 tiles against fp64
  std::vector<float> ya(X.size()), yb(X.size());
  hipLaunchKernelGGL(k_chain_f32, dim3(n_cu), dim3(WAVES * 64), lds_a, 0, dwa, db, dx, dy, tiles_per_wave, LAYERS);
  hipMemcpy(ya.data(), dy, X.size() * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL((k_chain_bf3), dim3(n_cu), dim3(WAVES * 64), lds_b, 0, dwb, db, dx, dy, tiles_per_wave, LAYERS);
  hipMemcpy(yb.data(), dy, X.size() * 4, hipMemcpyDeviceToHost);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  double ea = 0, eb = 0, scale = 0, eab = 0;
  for (long n = 0; n < 64 * 32; ++n) {
    double x[64], y[64];
    for (int f = 0; f < 64; ++f) x[f] = X[n * 64 + f];
    for (int l = 0; l < LAYERS; ++l) {
      for (int o = 0; o < 64; ++o) {
        double acc = B[l * 64 + o];
        for (int f = 0; f < 64; ++f) acc += (double)W[(l * 64 + o) * 64 + f] * x[f];
        y[o] = acc > 0 ? acc : 0;
      }
      for (int f = 0; f < 64; ++f) x[f] = y[f];
    }
    for (int f = 0; f < 64; ++f) {
      ea = fmax(ea, fabs(ya[n * 64 + f] - x[f]));
      eb = fmax(eb, fabs(yb[n * 64 + f] - x[f]));
      eab = fmax(eab, fabs((double)ya[n * 64 + f] - yb[n * 64 + f]));
      scale = fmax(scale, fabs(x[f]));
    }
  }
  printf("after %d layers, |y| up to %.3f:  max error vs fp64  fp32-MFMA %.3e   bf16x3 %.3e   (fp32 vs bf16x3 %.3e)\n", LAYERS, scale, ea, eb, eab);

  // ---- timing: 32 layers per tile
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lpt = 32;
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_chain_f32, dim3(n_cu), dim3(WAVES * 64), lds_a, 0, dwa, db, dx, dy, tiles_per_wave, lpt);
      else hipLaunchKernelGGL(k_chain_bf3, dim3(n_cu), dim3(WAVES * 64), lds_b, 0, dwb, db, dx, dy, tiles_per_wave, lpt);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = fminf(best, ms);
    }
    const double layer_tiles = (double)tiles * lpt, flop = layer_tiles * 2.0 * 64 * 64 * 32;
    printf("%-10s %.3f ms for %ld tiles x %d layers: %.0f ns per layer-tile per SIMD-slot, %.1f fp32-equivalent TFLOP/s\n",
           which == 0 ? "fp32 MFMA" : "bf16 x 3", best, tiles, lpt, best * 1e6 / (layer_tiles / (n_cu * 4.0)), flop / (best * 1e-3) / 1e12);
  }
  return 0;
}
