// bf16 VGG16 trunk of LPIPS on the matrix cores (reference utils/lpips/pretrained_networks.py:96-134 runs torchvision's
// fp32 convolutions through cuDNN; utils/lpips/lpips.py:81-123 is the head).  The trunk is 13 3x3 convolutions =
// 160 GFLOP per 512x512 image, 480 GFLOP per training step (two images forward, one backward): the only
// GEMM-shaped work of a GoMAvatar step that is large enough for MFMA, and ~97 % of the step's time when run through
// the library convolutions (scripts/bench_modes.py).
//
//   k_conv3x3_bf16      implicit GEMM, NHWC bf16 activations, fp32 accumulation in v_mfma_f32_16x16x32_bf16.
//                       Workgroup = 8x16 output pixels x 64 output channels, 4 waves (2 pixel rows x 64 channels
//                       each = 2x4 MFMA tiles).  Per 32-input-channel chunk the 10x18 halo patch and the 9x64x32
//                       weight block are staged in LDS once; the nine taps are shifted views of the same patch.
//                       A = weights [co][k], B = pixels [px][k]  ->  D[co][px]: every lane ends up with 4
//                       consecutive output channels of one pixel (one 8-byte NHWC store).
//                       Epilogue: + bias, ReLU, or (backward-data) * [forward activation > 0].
//                       The backward-data convolution is the same kernel on 180-degree-rotated, transposed weights.
//   k_maxpool2_*        2x2 max-pool forward / backward (argmax recomputed from the saved input)
//   k_lpips_prepare     (B,H,W,3) fp32 image in [0,1] -> ScalingLayer(2x-1) -> NHWC bf16 padded to 32 channels
//   k_lpips_head_nhwc_* the LPIPS head of lpips.hip for NHWC bf16 taps (one wave per pixel group)
#include "gom_internal.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // storage type

__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }

// ---- "bf16x3": fp32-accurate trunk on the bf16 matrix cores ---------------------------------------------------------------------
// Every activation, gradient and weight is kept as TWO bf16 planes, hi = bf16(v) and lo = bf16(v - hi) (16 mantissa bits together:
// 7.6e-6 relative), the lo plane `lo` elements behind the hi plane.  A convolution then runs over 3 x Cin/32 VIRTUAL input-channel
// chunks -- (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo) -- through the unchanged MFMA loop (fp32 accumulation; the dropped x_lo w_lo
// term is 2^-16 of a product), i.e. three v_mfma_f32_16x16x32_bf16 where the plain mode issues one: ~830 TFLOP/s of fp32-grade peak
// instead of the 157 TFLOP/s of v_mfma_f32_32x32x2_f32.  lo == 0 everywhere means the plain bf16 mode.
__device__ __forceinline__ void split_bf(float v, bf16_t &hi, bf16_t &lo) { hi = f2bf(v); lo = f2bf(v - bf2f(hi)); }
__device__ __forceinline__ void store4(bf16_t *p, size_t lo, const float (&v)[4]) {   // 4 consecutive channels: one 8-byte store per plane
    if (lo == 0) {
        uint2 o;
        o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2 *>(p) = o;
        return;
    }
    bf16_t h[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; r++) split_bf(v[r], h[r], l[r]);
    *reinterpret_cast<uint2 *>(p) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2 *>(p + lo) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}

// Which output channel an MFMA tile row stands for.  D[i][j] of tile n leaves lane (kg, l15) with rows 4 kg .. 4 kg + 3: with the natural order
// (row i of tile n = channel 16 n + i) a lane ends with NT separate groups of 4 channels -- NT 8-byte stores per pixel and plane, a quarter of a
// 32-byte sector each.  With row i of tile n = channel 4 NT (i / 4) + 4 n + i % 4 the same lane holds the 4 NT CONSECUTIVE channels
// 4 NT kg .. 4 NT kg + 4 NT - 1: one 32-byte store (NT = 4), and the four lanes of a pixel write its 128-byte line.  Only the row a weight fragment
// is fetched from changes (the staging kernels apply it to the SOURCE row, LDS rows stay consecutive: the swizzle's proof holds); the sums are the same.
template <int NT>
__device__ __forceinline__ int tile_row_channel(int i) { return 4 * NT * ((i & 15) >> 2) + 4 * (i >> 4) + (i & 3); }   // i = 16 n + row
// 4 NT consecutive channels of one pixel: 8 NT bytes per plane in 16-byte stores
template <int NT>
__device__ __forceinline__ void store_row(bf16_t *p, size_t lo, const float (&v)[4 * NT]) {
    uint32_t h[2 * NT], l[2 * NT];
#pragma unroll
    for (int k = 0; k < 2 * NT; k++) {
        bf16_t h0, h1, l0 = 0, l1 = 0;
        if (lo) { split_bf(v[2 * k], h0, l0); split_bf(v[2 * k + 1], h1, l1); } else { h0 = f2bf(v[2 * k]); h1 = f2bf(v[2 * k + 1]); }
        h[k] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        l[k] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
#pragma unroll
    for (int q = 0; q < NT / 2; q++) {
        *reinterpret_cast<uint4 *>(p + 8 * q) = make_uint4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
        if (lo) *reinterpret_cast<uint4 *>(p + lo + 8 * q) = make_uint4(l[4 * q], l[4 * q + 1], l[4 * q + 2], l[4 * q + 3]);
    }
}

constexpr int kTileW = 16, kBN = 64, kKC = 32;
constexpr int kPatchW = kTileW + 2;  // 18
// TH = pixel rows per workgroup (2 per wave): 8 rows / 4 waves for the small deep layers, 16 rows / 8 waves for the large
// ones -- the 36 KB weight block is re-read from L2 by every workgroup, so twice the pixels per workgroup halves that traffic
// (the large layers are L2-bandwidth-bound at 8 rows: 100 flop per byte staged).

// LDS rows are 64 bytes (32 channels of one pixel / one output channel) and a fragment read is one 16-byte k-group of 16
// consecutive rows per 16 lanes.  ds_read_b128 is served in four 16-lane groups over a 256-byte bank row, so with a 64-byte
// row stride the rows r and r + 4 of a group collide: 8 LDS cycles instead of 4 per instruction.  XOR-ing the k-group with 2
// for the rows whose bit 2 is set spreads every group over all 16 slots for ANY start row (the nine taps are shifted windows
// of the same patch; checked exhaustively over the lane groups of MI355X_MICROARCH.md).  LDS unit (row, slot) holds channel
// group slot ^ 2 * bit2(row): the writers apply it to the source (LDS-DMA writes linearly), the readers to the address.
__device__ __forceinline__ int swz_part(int part, int row) { return part ^ (((row >> 2) & 1) << 1); }

// Epilogue shared by the convolution kernels: the wave's RPW x 4 accumulator tiles -> + bias, ReLU, ReLU mask -> bf16 plane(s).
// D[i = co][j = px]: a lane holds co = 4 * kg + r (r = 0..3) of pixel column l15: one 8-byte NHWC store per plane.
// what a value reads back as from its stored plane(s): hi + lo (exact in fp32), or the one bf16
__device__ __forceinline__ float stored_value(float v, bool two_planes) {
    bf16_t h, l;
    if (!two_planes) return bf2f(f2bf(v));
    split_bf(v, h, l);
    return bf2f(h) + bf2f(l);
}

// `pooled` (forward layers in front of a 2 x 2 max-pool): the pooled tensor [B][H/2][W/2][Cout] is written here as well -- the maximum of the four STORED
// values (what k_maxpool2_fwd would read back; a wave's rows 2 w, 2 w + 1 are a lane's own, the neighbour column is lane ^ 1), so the pool's launch and its
// read of the full-resolution tensor disappear.  H, W even; tiles start at even coordinates.
template <bool RELU, int RPW>
__device__ __forceinline__ void conv_store(const f32x4 (&acc)[RPW][4], int H, int W, int Cout, size_t img, int ty0, int tx0, int row0, int co0, int l15, int kg,
                                           const float *__restrict__ bias, const bf16_t *__restrict__ mask, bf16_t *__restrict__ out, size_t out_lo,
                                           bf16_t *__restrict__ pooled, size_t pooled_lo) {
    const int co = co0 + 16 * kg;   // this lane's 16 consecutive channels (tile_row_channel<4>): 4 n + r <- acc[m][n][r]
    float keep[RPW][16];            // (read again only when pooled != null)
#pragma unroll
    for (int m = 0; m < RPW; m++) {
        const int gy = ty0 + row0 + m, gx = tx0 + l15;
        const bool in = gy < H && gx < W;
        const size_t pix = in ? (img + (size_t)gy * W + gx) * Cout : 0;
        float v[16];
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[4 * n + r] = acc[m][n][r];
        if (bias) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] += bias[co + k];
        }
        if (RELU) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = fmaxf(v[k], 0.f);
        }
        if (mask && in) {
            const uint4 m0 = *reinterpret_cast<const uint4 *>(mask + pix + co), m1 = *reinterpret_cast<const uint4 *>(mask + pix + co + 8);
            const uint32_t mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                v[2 * k] = bf2f((bf16_t)(mw[k] & 0xffff)) > 0.f ? v[2 * k] : 0.f;
                v[2 * k + 1] = bf2f((bf16_t)(mw[k] >> 16)) > 0.f ? v[2 * k + 1] : 0.f;
            }
        }
        if (in) store_row<4>(out + pix + co, out_lo, v);
#pragma unroll
        for (int k = 0; k < 16; k++) keep[m][k] = v[k];
    }
    if (pooled) {
#pragma unroll
        for (int m = 0; m + 1 < RPW; m += 2) {
            const int gy = ty0 + row0 + m, gx = tx0 + l15;
            float mx[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float a = fmaxf(stored_value(keep[m][k], out_lo != 0), stored_value(keep[m + 1][k], out_lo != 0));
                mx[k] = fmaxf(a, __shfl_xor(a, 1, 64));   // (every lane takes part; the stores below are the conditional part)
            }
            if (!(l15 & 1) && gy < H && gx < W) store_row<4>(pooled + (img / 4 + (size_t)(gy >> 1) * (W >> 1) + (size_t)(gx >> 1)) * Cout + co, pooled_lo, mx);
        }
    }
}

// Split-K: raw fp32 partial sums of this workgroup's share of the input channels -> partial [splits][B][H][W][Cout]; k_splitk_epilogue
// adds the shares up in split order and applies bias / ReLU / mask.  (Summing inside the convolution, in the tile's last workgroup, was
// built in round 4 and is slower: LABBOOK R4.7.)
template <int RPW>
__device__ __forceinline__ void conv_store_partial(const f32x4 (&acc)[RPW][4], int H, int W, int Cout, size_t img, int nb, int zs, int ty0, int tx0, int row0, int co0,
                                                   int l15, int kg, float *__restrict__ partial) {
#pragma unroll
    for (int m = 0; m < RPW; m++) {
        const int gy = ty0 + row0 + m, gx = tx0 + l15;
        if (gy >= H || gx >= W) continue;
        float *dst = partial + (size_t)zs * nb * H * W * Cout + (img + (size_t)gy * W + gx) * Cout + co0 + 16 * kg;   // 16 consecutive channels (tile_row_channel<4>)
#pragma unroll
        for (int n = 0; n < 4; n++) *reinterpret_cast<f32x4 *>(dst + 4 * n) = acc[m][n];
    }
}

// in  [B][H][W][Cin]  bf16 (Cin multiple of 32);  wt [Cin/32][9][Cout][32] bf16 (Cout multiple of 64)
// out [B][H][W][Cout] bf16;  bias fp32 [Cout] or null;  mask (same shape as out) or null: out *= (mask > 0)
// SPLITK: blockIdx.z = image * splits + s; this block sums only its share of the input-channel chunks and stores raw fp32
// partial sums to `partial` [splits][B][H][W][Cout]; k_splitk_epilogue adds them up and applies bias / ReLU / mask.  Used
// for the deep layers (64x64 and 32x32 pixels: 8-32 pixel tiles), which would otherwise leave most of the 256 CUs idle.
template <bool RELU, bool SPLITK, int TH>
__global__ void __launch_bounds__(TH * 32) k_conv3x3_bf16(int H, int W, int Cin, int Cout, const bf16_t *__restrict__ in,
                                                      const bf16_t *__restrict__ wt, const float *__restrict__ bias,
                                                      const bf16_t *__restrict__ mask, bf16_t *__restrict__ out, int splits,
                                                      float *__restrict__ partial, size_t in_lo, size_t out_lo, bf16_t *__restrict__ pooled, size_t pooled_lo) {
    constexpr int kTileH = TH, kPatchPx = (TH + 2) * kPatchW, NT = TH * 32;
    __shared__ __attribute__((aligned(16))) bf16_t s_in[kPatchPx * kKC];   // 11 520 B (TH = 8) / 20 736 B (TH = 16)
    __shared__ __attribute__((aligned(16))) bf16_t s_w[9 * kBN * kKC];      // 36 864 B
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int tx0 = (blockIdx.x % tiles_x) * kTileW, ty0 = (blockIdx.x / tiles_x) * kTileH;
    const int co0 = blockIdx.y * kBN;
    const int zb = SPLITK ? (int)blockIdx.z / splits : (int)blockIdx.z, zs = SPLITK ? (int)blockIdx.z % splits : 0;
    const size_t img = (size_t)zb * H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunk = in_lo ? 3 * (Cin / kKC) : Cin / kKC;   // bf16x3: three virtual chunks per 32 input channels
    const int cc_lo = SPLITK ? zs * nchunk / splits : 0, cc_hi = SPLITK ? (zs + 1) * nchunk / splits : nchunk;
    // (A register-prefetch software pipeline -- next chunk's global loads in flight during the MFMA loop -- was measured
    //  25 % SLOWER: +50 VGPRs, spills and one wave less per SIMD; latency is hidden by the 2-3 co-resident workgroups instead.)
    for (int cc = cc_lo; cc < cc_hi; cc++) {
        __syncthreads();  // the previous chunk's MFMA reads are done
        const int sc = in_lo ? cc / 3 : cc;                                   // source chunk; the x_lo plane for the middle one of a triple
        const bf16_t *src_plane = in + ((in_lo && cc % 3 == 1) ? in_lo : 0);
        for (int idx = tid; idx < kPatchPx * 4; idx += NT) {
            const int px = idx >> 2, part = idx & 3;
            const int gy = ty0 + px / kPatchW - 1, gx = tx0 + px % kPatchW - 1;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const uint4 *>(src_plane + (img + (size_t)gy * W + gx) * Cin + sc * kKC + part * 8);
            *reinterpret_cast<uint4 *>(s_in + px * kKC + swz_part(part, px) * 8) = v;
        }
        {
            const bf16_t *wsrc = wt + ((size_t)cc * 9 * Cout + co0) * kKC;
            for (int idx = tid; idx < 9 * kBN * 4; idx += NT) {
                const int tap = idx >> 8, r = idx & 255;  // 256 16-byte units per tap (64 co x 32 ci)
                *reinterpret_cast<uint4 *>(s_w + tap * kBN * kKC + (r >> 2) * kKC + swz_part(r & 3, r >> 2) * 8) =
                    *reinterpret_cast<const uint4 *>(wsrc + (size_t)tap * Cout * kKC + tile_row_channel<4>(r >> 2) * kKC + (r & 3) * 8);   // LDS row r >> 2 <- its channel's weights
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
            bf16x8 bfrag[2], afrag[4];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int px = (2 * wave + m + ky) * kPatchW + l15 + kx;
                bfrag[m] = *reinterpret_cast<const bf16x8 *>(s_in + px * kKC + swz_part(kg, px) * 8);
            }
#pragma unroll
            for (int n = 0; n < 4; n++)
                afrag[n] = *reinterpret_cast<const bf16x8 *>(s_w + (tap * kBN + n * 16 + l15) * kKC + swz_part(kg, l15) * 8);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[n], bfrag[m], acc[m][n], 0, 0, 0);
        }
    }
    if (SPLITK) conv_store_partial<2>(acc, H, W, Cout, img, (int)gridDim.z / splits, zs, ty0, tx0, 2 * wave, co0, l15, kg, partial);
    else conv_store<RELU, 2>(acc, H, W, Cout, img, ty0, tx0, 2 * wave, co0, l15, kg, bias, mask, out, out_lo, pooled, pooled_lo);
}

// ---- pipelined variant: global -> LDS by LDS-DMA (global_load_lds_dwordx4), three stages in flight ---------------------
// Same tile (16 x 16 pixels x 64 output channels, 8 waves of 2 rows), same MFMA / epilogue as k_conv3x3_bf16<.., 16>; what
// changes is how the operands reach LDS.  v1 stages a whole 32-channel chunk (halo patch + 9 taps of weights, 57 KB) through
// VGPRs between two barriers and relies on a second co-resident workgroup to keep the matrix cores busy meanwhile.  Here a
// STAGE is one kernel row (ky) of one chunk: 3 taps of weights (12 KB) and, for ky = 0, the chunk's halo patch (20 KB).  Every
// wave issues the loads of stage s + 2 as `global_load_lds` (no VGPRs, no ds_write pass: lane i's 16 bytes land at
// base + 16 i, so the LDS images are laid out in exactly the order the lanes enumerate them), waits with a COUNTED
// `s_waitcnt vmcnt(N)` that leaves the younger stages' loads in flight, crosses ONE barrier and runs the 24 MFMAs of stage
// s.  Weights are triple-buffered, patches double-buffered: 78 KB, two workgroups per CU.
//   * Halo pixels outside the image read 16 zero bytes from a global constant instead of being masked: every wave issues
//     the same number of loads in every stage, which is what makes the counted waits exact.
//   * RAW: a wave reads a buffer only after its own vmcnt wait AND the barrier every other wave reaches after theirs.
//     WAR: the loads that overwrite a buffer are issued after the barrier that follows its last reader's MFMAs.
//   * No ordinary global load may sit inside the loop (hipcc would drain the DMA queue with vmcnt(0) at its first use).
__device__ uint4 g_zero16[4];   // zero-initialised: source of the halo's out-of-image pixels

constexpr int kV2PatchUnits = 18 * 18 * 4;            // 16-byte units of a halo patch (1 296)
constexpr int kV2PatchBytes = kV2PatchUnits * 16;     // 20 736
constexpr int kV2WBytes = 3 * kBN * kKC * 2;          // 12 288: three taps
constexpr int kV2Lds = 2 * kV2PatchBytes + 3 * kV2WBytes;   // 78 336

// Issued as inline assembly, not through __builtin_amdgcn_global_load_lds: the compiler's wait-count pass books the builtin as a FLAT
// access that may touch LDS, and while one is pending every LDS wait it inserts is lgkmcnt(0) -- with DMA loads always in flight the
// fragment reads of the next tap could never stay outstanding behind the MFMAs of the current one.  The loop's vmcnt waits are
// written by hand anyway (counted, see below).
__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    const uint32_t m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0) : "memory", "m0");
}

template <bool RELU, bool SPLITK, int RPW>
__global__ void __launch_bounds__(1024 / RPW, 2) k_conv3x3_bf16_v2(int H, int W, int Cin, int Cout, const bf16_t *__restrict__ in,
                                                                const bf16_t *__restrict__ wt, const float *__restrict__ bias,
                                                                const bf16_t *__restrict__ mask, bf16_t *__restrict__ out, int splits,
                                                                float *__restrict__ partial, size_t in_lo, size_t out_lo, bf16_t *__restrict__ pooled, size_t pooled_lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TH = 16, NW = TH / RPW;                                  // RPW pixel rows per wave, NW waves
    constexpr int PU = kV2PatchUnits / NW, WU = 3 * kBN * 4 / NW;          // 16-byte units per wave: patch (162 / 324), weights (96 / 192)
    constexpr int PI = (PU + 63) / 64, WI = (WU + 63) / 64;                // load instructions per wave
    static_assert(kV2PatchUnits % NW == 0 && (3 * kBN * 4) % NW == 0, "even shares");
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int tx0 = (blockIdx.x % tiles_x) * kTileW, ty0 = (blockIdx.x / tiles_x) * TH;
    const int co0 = blockIdx.y * kBN;
    const int zb = SPLITK ? (int)blockIdx.z / splits : (int)blockIdx.z, zs = SPLITK ? (int)blockIdx.z % splits : 0;
    const size_t img = (size_t)zb * H * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int nchunk = in_lo ? 3 * (Cin / kKC) : Cin / kKC;   // bf16x3: three virtual chunks per 32 input channels
    const int cc_lo = SPLITK ? zs * nchunk / splits : 0, cc_hi = SPLITK ? (zs + 1) * nchunk / splits : nchunk;
    const int NS = 3 * (cc_hi - cc_lo);   // stages

    // this wave's share of a patch: units [PU w, PU w + PU); of a weight stage: [WU w, WU w + WU); 64 lanes per instruction
    const unsigned char *zero = reinterpret_cast<const unsigned char *>(g_zero16);
    const unsigned char *psrc[PI];   // element 0 of the pixel's 32-channel row in chunk 0 (or the zero block)
    bool pin[PI];
#pragma unroll
    for (int j = 0; j < PI; j++) {
        const int u = min(PU * wave + 64 * j + lane, kV2PatchUnits - 1);
        const int px = u >> 2, part = swz_part(u & 3, px);   // LDS unit u holds channel group `part` of its pixel (see swz_part)
        const int gy = ty0 + px / kPatchW - 1, gx = tx0 + px % kPatchW - 1;
        pin[j] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        psrc[j] = pin[j] ? reinterpret_cast<const unsigned char *>(in + (img + (size_t)gy * W + gx) * Cin + part * 8) : zero;
    }
    // weight units of a stage: v = WU w + 64 j + lane -> tap v >> 8 (of the row), r = v & 255
    size_t woff[WI];
#pragma unroll
    for (int j = 0; j < WI; j++) {
        const int v = min(WU * wave + 64 * j + lane, 3 * kBN * 4 - 1);
        const int row = (v & 255) >> 2;                                           // output channel of the unit
        woff[j] = ((size_t)(v >> 8) * Cout * kKC + (size_t)tile_row_channel<4>(row) * kKC + (size_t)swz_part(v & 3, row) * 8) * 2;   // bytes inside the (chunk, ky) row block: LDS row <- its channel (tile_row_channel)
    }
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wt + (size_t)co0 * kKC);

    auto issue = [&](int s) {   // loads of stage s (s < NS): WI weight instructions, + PI patch instructions when ky == 0
        const int cc = cc_lo + s / 3, ky = s % 3;
        unsigned char *wb = smem + 2 * kV2PatchBytes + (s % 3) * kV2WBytes + (WU * wave) * 16;
        const unsigned char *wsrc = wbase + ((size_t)(cc * 9 + ky * 3) * Cout * kKC) * 2;
#pragma unroll
        for (int j = 0; j < WI; j++)
            if (64 * j + lane < WU) glds16(wsrc + woff[j], wb + 64 * j * 16);
        if (ky == 0) {
            unsigned char *pb = smem + ((s / 3) & 1) * kV2PatchBytes + (PU * wave) * 16;
            const size_t coff = in_lo ? (size_t)(cc / 3) * kKC * 2 + (cc % 3 == 1 ? in_lo * 2 : 0) : (size_t)cc * kKC * 2;   // (virtual chunk -> source chunk and plane)
#pragma unroll
            for (int j = 0; j < PI; j++)
                if (64 * j + lane < PU) glds16(pin[j] ? psrc[j] + coff : psrc[j], pb + 64 * j * 16);
        }
    };

    f32x4 acc[RPW][4];
#pragma unroll
    for (int m = 0; m < RPW; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Every scalar argument the epilogue needs is pulled into SGPRs HERE: a scalar load left pending across the loop makes lgkmcnt count two
    // kinds of events, and the wait-count pass then turns every LDS wait of the loop into lgkmcnt(0).
    asm volatile("" ::"s"(bias), "s"(mask), "s"(out), "s"(partial), "s"(out_lo), "s"(Cout), "s"(splits), "s"(pooled), "s"(pooled_lo));
    issue(0);
    if (NS > 1) issue(1);
    for (int s = 0; s < NS; s++) {
        // stage s has landed when at most the loads of stage s + 1 are still in flight (WI, or WI + PI when it carries a patch)
        if (s + 1 >= NS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if ((s + 1) % 3 == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WI + PI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WI) : "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < NS) issue(s + 2);
        const int ky = s % 3;
        const bf16_t *pb = reinterpret_cast<const bf16_t *>(smem + ((s / 3) & 1) * kV2PatchBytes);
        const bf16_t *wb = reinterpret_cast<const bf16_t *>(smem + 2 * kV2PatchBytes + (s % 3) * kV2WBytes);
        // Fragments of tap kx + 1 are requested BEFORE the MFMAs of tap kx issue (two register sets, counted lgkmcnt waits).  Measured
        // neutral against one set re-filled behind the 5th MFMA (what hipcc schedules by itself): with four waves per SIMD another wave's
        // MFMAs cover the read latency either way (LABBOOK R4.7); kept because it no longer depends on that occupancy.
        bf16x8 bfrag[2][RPW], afrag[2][4];
        auto fetch = [&](int set, int kx) {
#pragma unroll
            for (int m = 0; m < RPW; m++) {
                const int px = (RPW * wave + m + ky) * kPatchW + l15 + kx;
                bfrag[set][m] = *reinterpret_cast<const bf16x8 *>(pb + px * kKC + swz_part(kg, px) * 8);
            }
#pragma unroll
            for (int n = 0; n < 4; n++)
                afrag[set][n] = *reinterpret_cast<const bf16x8 *>(wb + (kx * kBN + n * 16 + l15) * kKC + swz_part(kg, l15) * 8);
        };
        fetch(0, 0);
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            if (kx < 2) fetch((kx + 1) & 1, kx + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < RPW; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[kx & 1][n], bfrag[kx & 1][m], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (SPLITK) conv_store_partial<RPW>(acc, H, W, Cout, img, (int)gridDim.z / splits, zs, ty0, tx0, RPW * wave, co0, l15, kg, partial);
    else conv_store<RELU, RPW>(acc, H, W, Cout, img, ty0, tx0, RPW * wave, co0, l15, kg, bias, mask, out, out_lo, pooled, pooled_lo);
}

// ---- bf16x3 with SHARED stages (round 5): one stage = one kernel row of one 32-channel GROUP, all three products ----------------
// k_conv3x3_bf16_v2 walks bf16x3 as three virtual chunks per 32 input channels -- (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo) -- each with its
// own patch and its own weight stages: x_hi and w_hi cross the LDS twice, and every MFMA pays 0.75 fragment reads (RPW + 4 ds_read_b128 per
// 4 RPW MFMAs).  The LDS pipe is what the K loop of that kernel runs against: per stage and CU 288 fragment reads x 1 KB + 38 KB of DMA
// writes against 1 536 MFMA cycles per SIMD (LABBOOK R5.3).  Here a stage carries w_hi AND w_lo of a kernel row (24 KB) and both patches of
// the group stay resident (x_hi, x_lo: 41 KB, double-buffered per group): per tap 4 A_hi + 4 A_lo + RPW B_hi + RPW B_lo fragment reads feed
// 12 RPW MFMAs -- 0.5 reads per MFMA -- and a third less DMA.  157 KB of LDS: ONE workgroup of 8 waves per CU (the loop no longer leans on a
// second workgroup: 72 MFMAs per wave between barriers, fragments of the next tap in flight behind the current tap's).
// Same weights buffer as v2 (chunks 3g and 3g + 2 of lpips.pack_conv_weight_x3; chunk 3g + 1, the duplicate of w_hi, is not read).
// The accumulation order differs from v2's (per tap hi.hi, lo.hi, hi.lo instead of three passes over the taps): fp32 round-off apart.
constexpr int kX3WBytes = 2 * kV2WBytes;                              // 24 576: three taps of w_hi, then three taps of w_lo
constexpr int kX3Lds = 4 * kV2PatchBytes + 3 * kX3WBytes;             // 156 672

template <bool RELU, bool SPLITK, int RPW>
__global__ void __launch_bounds__(1024 / RPW, 1) k_conv3x3_x3s(int H, int W, int Cin, int Cout, const bf16_t *__restrict__ in, const bf16_t *__restrict__ wt,
                                                          const float *__restrict__ bias, const bf16_t *__restrict__ mask, bf16_t *__restrict__ out, int splits,
                                                          float *__restrict__ partial, size_t in_lo, size_t out_lo, bf16_t *__restrict__ pooled, size_t pooled_lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TH = 16, NW = TH / RPW;                                  // RPW pixel rows per wave (2: 8 waves; 4: 4 waves with 0.33 fragment reads per MFMA)
    constexpr int PU = kV2PatchUnits / NW, WU = 3 * kBN * 4 / NW;          // 16-byte units per wave: a patch plane (162 / 324), a weight plane of a stage (96 / 192)
    constexpr int PI = (PU + 63) / 64, WI = (WU + 63) / 64;                // load instructions per wave and plane (3, 2 / 6, 3)
    const int tiles_x = (W + kTileW - 1) / kTileW;
    const int tx0 = (blockIdx.x % tiles_x) * kTileW, ty0 = (blockIdx.x / tiles_x) * TH;
    const int co0 = blockIdx.y * kBN;
    const int zb = SPLITK ? (int)blockIdx.z / splits : (int)blockIdx.z, zs = SPLITK ? (int)blockIdx.z % splits : 0;
    const size_t img = (size_t)zb * H * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int G = Cin / kKC;                                               // 32-channel groups
    const int g_lo = SPLITK ? zs * G / splits : 0, g_hi = SPLITK ? (zs + 1) * G / splits : G;
    const int NS = 3 * (g_hi - g_lo);   // stages: (group, ky)

    const unsigned char *zero = reinterpret_cast<const unsigned char *>(g_zero16);
    const unsigned char *psrc[PI];
    bool pin[PI];
#pragma unroll
    for (int j = 0; j < PI; j++) {
        const int u = min(PU * wave + 64 * j + lane, kV2PatchUnits - 1);
        const int px = u >> 2, part = swz_part(u & 3, px);
        const int gy = ty0 + px / kPatchW - 1, gx = tx0 + px % kPatchW - 1;
        pin[j] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        psrc[j] = pin[j] ? reinterpret_cast<const unsigned char *>(in + (img + (size_t)gy * W + gx) * Cin + part * 8) : zero;
    }
    size_t woff[WI];
#pragma unroll
    for (int j = 0; j < WI; j++) {
        const int v = min(WU * wave + 64 * j + lane, 3 * kBN * 4 - 1);
        const int row = (v & 255) >> 2;
        woff[j] = ((size_t)(v >> 8) * Cout * kKC + (size_t)tile_row_channel<4>(row) * kKC + (size_t)swz_part(v & 3, row) * 8) * 2;
    }
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(wt + (size_t)co0 * kKC);

    auto issue = [&](int s) {   // loads of stage s: 2 WI weight instructions (w_hi, w_lo), + 2 PI patch instructions (x_hi, x_lo) when ky == 0
        const int g = g_lo + s / 3, ky = s % 3;
        unsigned char *wb = smem + 4 * kV2PatchBytes + (s % 3) * kX3WBytes + (WU * wave) * 16;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {   // virtual chunks 3 g (w_hi) and 3 g + 2 (w_lo) of the packed weights
            const unsigned char *wsrc = wbase + ((size_t)((3 * g + 2 * pl) * 9 + ky * 3) * Cout * kKC) * 2;
#pragma unroll
            for (int j = 0; j < WI; j++)
                if (64 * j + lane < WU) glds16(wsrc + woff[j], wb + pl * kV2WBytes + 64 * j * 16);
        }
        if (ky == 0) {
            unsigned char *pb = smem + ((s / 3) & 1) * 2 * kV2PatchBytes + (PU * wave) * 16;
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                const size_t coff = (size_t)g * kKC * 2 + (pl ? in_lo * 2 : 0);
#pragma unroll
                for (int j = 0; j < PI; j++)
                    if (64 * j + lane < PU) glds16(pin[j] ? psrc[j] + coff : psrc[j], pb + pl * kV2PatchBytes + 64 * j * 16);
            }
        }
    };

    f32x4 acc[RPW][4];
#pragma unroll
    for (int m = 0; m < RPW; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    asm volatile("" ::"s"(bias), "s"(mask), "s"(out), "s"(partial), "s"(out_lo), "s"(Cout), "s"(splits), "s"(pooled), "s"(pooled_lo));   // (see k_conv3x3_bf16_v2)
    issue(0);
    if (NS > 1) issue(1);
    for (int s = 0; s < NS; s++) {
        if (s + 1 >= NS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if ((s + 1) % 3 == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WI + 2 * PI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WI) : "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < NS) issue(s + 2);
        const int ky = s % 3;
        const bf16_t *pbh = reinterpret_cast<const bf16_t *>(smem + ((s / 3) & 1) * 2 * kV2PatchBytes);
        const bf16_t *pbl = reinterpret_cast<const bf16_t *>(smem + ((s / 3) & 1) * 2 * kV2PatchBytes + kV2PatchBytes);
        const bf16_t *wbh = reinterpret_cast<const bf16_t *>(smem + 4 * kV2PatchBytes + (s % 3) * kX3WBytes);
        const bf16_t *wbl = reinterpret_cast<const bf16_t *>(smem + 4 * kV2PatchBytes + (s % 3) * kX3WBytes + kV2WBytes);
        bf16x8 bh[2][RPW], bl[2][RPW], ah[2][4], al[2][4];
        auto fetch = [&](int set, int kx) {
#pragma unroll
            for (int m = 0; m < RPW; m++) {
                const int px = (RPW * wave + m + ky) * kPatchW + l15 + kx;
                const int o = px * kKC + swz_part(kg, px) * 8;
                bh[set][m] = *reinterpret_cast<const bf16x8 *>(pbh + o);
                bl[set][m] = *reinterpret_cast<const bf16x8 *>(pbl + o);
            }
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const int o = (kx * kBN + n * 16 + l15) * kKC + swz_part(kg, l15) * 8;
                ah[set][n] = *reinterpret_cast<const bf16x8 *>(wbh + o);
                al[set][n] = *reinterpret_cast<const bf16x8 *>(wbl + o);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            if (kx < 2) fetch((kx + 1) & 1, kx + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int c = kx & 1;
#pragma unroll
            for (int m = 0; m < RPW; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c][n], bh[c][m], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RPW; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c][n], bl[c][m], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RPW; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[c][n], bh[c][m], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (SPLITK) conv_store_partial<RPW>(acc, H, W, Cout, img, (int)gridDim.z / splits, zs, ty0, tx0, RPW * wave, co0, l15, kg, partial);
    else conv_store<RELU, RPW>(acc, H, W, Cout, img, ty0, tx0, RPW * wave, co0, l15, kg, bias, mask, out, out_lo, pooled, pooled_lo);
}

__device__ __forceinline__ void unpack8(const uint4 q, float (&f)[8]) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) { f[2 * k] = bf2f((bf16_t)(w[k] & 0xffff)); f[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16)); }
}
// 8 consecutive channels of a tensor kept as one or two bf16 planes
__device__ __forceinline__ void load8(const bf16_t *p, size_t lo, float (&f)[8]) {
    unpack8(*reinterpret_cast<const uint4 *>(p), f);
    if (lo) {
        float l[8];
        unpack8(*reinterpret_cast<const uint4 *>(p + lo), l);
#pragma unroll
        for (int k = 0; k < 8; k++) f[k] += l[k];
    }
}
__device__ __forceinline__ float load1(const bf16_t *p, size_t lo) { return lo ? bf2f(p[0]) + bf2f(p[lo]) : bf2f(p[0]); }
__device__ __forceinline__ void store1(bf16_t *p, size_t lo, float v) {
    if (lo) { bf16_t h, l; split_bf(v, h, l); p[0] = h; p[lo] = l; } else p[0] = f2bf(v);
}
__device__ __forceinline__ void store8(bf16_t *p, size_t lo, const float (&v)[8]) { store_row<2>(p, lo, v); }   // 8 consecutive channels: one 16-byte store per plane

// sum of the split-K partials + bias, ReLU, mask -> bf16; 8 channels per thread (16-byte stores per plane).  POOL: a thread owns a 2 x 2 window's four
// pixels and writes their maximum (of the stored values) to `pooled` as well, like conv_store.
template <bool POOL>
__global__ void __launch_bounds__(256) k_splitk_epilogue(size_t n8, int H, int W, int Cout, int splits, const float *__restrict__ partial,
                                                         const float *__restrict__ bias, const bf16_t *__restrict__ mask, bf16_t *__restrict__ out,
                                                         int relu, size_t out_lo, bf16_t *__restrict__ pooled, size_t pooled_lo) {
    const size_t stride = n8 * 8;
    const int C8 = Cout / 8;
    auto one = [&](size_t i, float (&v)[8]) {   // i: (pixel, group of 8 channels)
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(partial + 8 * i), a1 = *reinterpret_cast<const f32x4 *>(partial + 8 * i + 4);
        for (int s = 1; s < splits; s++) {
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(partial + (size_t)s * stride + 8 * i), b1 = *reinterpret_cast<const f32x4 *>(partial + (size_t)s * stride + 8 * i + 4);
            a0[0] += b0[0]; a0[1] += b0[1]; a0[2] += b0[2]; a0[3] += b0[3];
            a1[0] += b1[0]; a1[1] += b1[1]; a1[2] += b1[2]; a1[3] += b1[3];
        }
        const int co = (int)((8 * i) % Cout);
        v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3]; v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
        if (bias) {
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] += bias[co + r];
        }
        if (relu) {
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = fmaxf(v[r], 0.f);
        }
        if (mask) {
            float mk[8];
            unpack8(*reinterpret_cast<const uint4 *>(mask + 8 * i), mk);
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = mk[r] > 0.f ? v[r] : 0.f;
        }
        store8(out + 8 * i, out_lo, v);
    };
    if constexpr (POOL) {
        const int Ho = H / 2, Wo = W / 2;
        const size_t nwin = n8 / 4;   // (image, window, group of 8 channels)
        for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < nwin; j += (size_t)gridDim.x * 256) {
            const int c8 = (int)(j % C8);
            const size_t wv = j / C8;
            const int xo = (int)(wv % Wo), yo = (int)((wv / Wo) % Ho);
            const size_t b = wv / ((size_t)Wo * Ho);
            const size_t p0 = ((size_t)b * H + 2 * yo) * W + 2 * xo;
            float mx[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float v[8];
                one((p0 + (size_t)(q >> 1) * W + (q & 1)) * C8 + c8, v);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float sv = stored_value(v[r], out_lo != 0);
                    mx[r] = q == 0 ? sv : fmaxf(mx[r], sv);
                }
            }
            store8(pooled + (wv * C8 + c8) * 8, pooled_lo, mx);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
            float v[8];
            one(i, v);
        }
    }
}

// ---- 2x2 max-pool (NHWC bf16), 8 channels per thread ------------------------------------------------------------
__global__ void __launch_bounds__(256) k_maxpool2_fwd(int B, int H, int W, int C, const bf16_t *__restrict__ x, bf16_t *__restrict__ y, size_t x_lo, size_t y_lo) {
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const size_t total = (size_t)B * Ho * Wo * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        const size_t p = i / C8;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), b = (int)(p / ((size_t)Wo * Ho));
        const bf16_t *src = x + (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + c8 * 8;
        const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
        float m[8];
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4 *>(src + off[w4]), f);
            if (x_lo) {   // bf16x3: the value is hi + lo (exact in fp32)
                float l[8];
                unpack8(*reinterpret_cast<const uint4 *>(src + x_lo + off[w4]), l);
#pragma unroll
                for (int k = 0; k < 8; k++) f[k] += l[k];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) m[k] = w4 == 0 ? f[k] : fmaxf(m[k], f[k]);
        }
        bf16_t *dst = y + (((size_t)b * Ho + yo) * Wo + xo) * C + c8 * 8;
        store8(dst, y_lo, m);   // (a maximum is one of the inputs: hi + lo represents it exactly again)
    }
}

// dx (+)= dy routed to the first maximum of each window (row-major order, like the reference framework), times the
// ReLU derivative [x > 0] of the layer that produced x (x = the pool's input = a post-ReLU activation);
// accumulate: dx already holds another gradient (the LPIPS head's) for this activation
__global__ void __launch_bounds__(256) k_maxpool2_bwd(int B, int H, int W, int C, const bf16_t *__restrict__ x, const bf16_t *__restrict__ dy,
                                                      bf16_t *__restrict__ dx, int accumulate, size_t x_lo, size_t dy_lo, size_t dx_lo) {
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;   // 8 channels per thread: 16-byte loads and stores (the kernel is HBM-bound: x, dy, dx twice when it accumulates)
    const size_t total = (size_t)B * Ho * Wo * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        const size_t p = i / C8;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), b = (int)(p / ((size_t)Wo * Ho));
        const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + c8 * 8;
        const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
        float v[4][8], g[8];
#pragma unroll
        for (int k = 0; k < 4; k++) load8(x + base + off[k], x_lo, v[k]);
        load8(dy + p * C + c8 * 8, dy_lo, g);
        int am[8];      // first maximum of each channel's window in row-major order, like the reference framework
        bool pos[8];    // ... and the ReLU derivative of the layer that produced it
#pragma unroll
        for (int ch = 0; ch < 8; ch++) {
            float best = v[0][ch];
            am[ch] = 0;
#pragma unroll
            for (int q = 1; q < 4; q++)
                if (v[q][ch] > best) { best = v[q][ch]; am[ch] = q; }
            pos[ch] = best > 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float o[8];
            if (accumulate) load8(dx + base + off[k], dx_lo, o);
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                const float add = (k == am[ch] && pos[ch]) ? g[ch] : 0.f;
                o[ch] = accumulate ? o[ch] + add : add;
            }
            store8(dx + base + off[k], dx_lo, o);
        }
    }
}

// ---- image -> trunk input: ((2x - 1) - shift) / scale, NHWC bf16 padded to 32 channels ------------------------------
__global__ void __launch_bounds__(256) k_lpips_prepare(size_t npix, const float *__restrict__ rgb, bf16_t *__restrict__ out, size_t out_lo) {
    const float shift[3] = {-0.030f, -0.088f, -0.188f}, scale[3] = {0.458f, 0.448f, 0.450f};
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
        bf16_t c[3], cl[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = ((2.f * rgb[3 * p + k] - 1.f) - shift[k]) / scale[k];
            if (out_lo) split_bf(v, c[k], cl[k]); else c[k] = f2bf(v);
        }
        for (int plane = 0; plane < (out_lo ? 2 : 1); plane++) {
            const bf16_t *q = plane ? cl : c;
            uint4 *dst = reinterpret_cast<uint4 *>(out + (plane ? out_lo : 0) + p * 32);
            dst[0] = make_uint4((uint32_t)q[0] | ((uint32_t)q[1] << 16), (uint32_t)q[2], 0u, 0u);
#pragma unroll
            for (int k = 1; k < 4; k++) dst[k] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}
// gradient wrt the (B,H,W,3) image in [0,1] from the gradient wrt the trunk input (first 3 of Cpad channels)
__global__ void __launch_bounds__(256) k_lpips_unprepare(size_t npix, int Cpad, const bf16_t *__restrict__ d_in, float *__restrict__ d_rgb, size_t in_lo) {
    const float scale[3] = {0.458f, 0.448f, 0.450f};
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 3; k++) d_rgb[3 * p + k] = load1(d_in + p * Cpad + k, in_lo) * (2.f / scale[k]);
    }
}

// ---- the trunk's FIRST layer without its channel padding (round 4) -----------------------------------------------------------------------
// conv1_1 has 3 input channels; padded to the 32-channel chunk of k_conv3x3_bf16 it issues nine taps of 32 channels of which 29 are zero
// (9.7 GFLOP per 512^2 image instead of 0.9), and its backward-data pass -- 64 -> 3 channels padded to 64 -- is a full 64 -> 64 convolution.
// Here the input is laid out as IM2COL rows instead: channel k = 3 (3 ky + kx) + c of pixel p holds the scaled value of pixel p + (ky-1, kx-1),
// zero outside the image (27 of 32 channels used), so that conv1_1 is a 1 x 1 convolution with K = 32 -- one MFMA k-step per plane pair --
// and its backward-data pass a 1 x 1 convolution 64 -> 32 followed by the col2im gather inside the "unprepare" kernel.  Both are HBM-bound
// streams (forward: 128 B in, 256 B out per pixel and plane pair) instead of MFMA-bound tiles.  Same products, same fp32 accumulation.
__global__ void __launch_bounds__(256) k_lpips_prepare_im2col(int B, int H, int W, const float *__restrict__ rgb, bf16_t *__restrict__ out, size_t out_lo) {
    const float shift[3] = {-0.030f, -0.088f, -0.188f}, scale[3] = {0.458f, 0.448f, 0.450f};
    const size_t npix = (size_t)B * H * W;
    // one thread per (pixel, group of 8 channels): a 16-byte store per plane
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix * 4; i += (size_t)gridDim.x * 256) {
        const size_t p = i >> 2;
        const int part = (int)(i & 3);
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const size_t img0 = p - (size_t)y * W - x;   // first pixel of the image
        bf16_t hi[8], lo[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int k = part * 8 + r;
            float v = 0.f;
            if (k < 27) {
                const int tap = k / 3, c = k % 3;
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = ((2.f * rgb[3 * (img0 + (size_t)yy * W + xx) + c] - 1.f) - shift[c]) / scale[c];
            }
            if (out_lo) split_bf(v, hi[r], lo[r]); else { hi[r] = f2bf(v); lo[r] = 0; }
        }
        auto pack = [](const bf16_t (&q)[8]) { return make_uint4((uint32_t)q[0] | ((uint32_t)q[1] << 16), (uint32_t)q[2] | ((uint32_t)q[3] << 16),
                                                                   (uint32_t)q[4] | ((uint32_t)q[5] << 16), (uint32_t)q[6] | ((uint32_t)q[7] << 16)); };
        *reinterpret_cast<uint4 *>(out + p * 32 + part * 8) = pack(hi);
        if (out_lo) *reinterpret_cast<uint4 *>(out + out_lo + p * 32 + part * 8) = pack(lo);
    }
}
// d rgb of the (B,H,W,3) image in [0,1] from the gradient w.r.t. the im2col rows: pixel p's channel c collects column 3 t + c of every pixel
// q = p - offset(t) that has p as its tap t
__global__ void __launch_bounds__(256) k_lpips_unprepare_col2im(int B, int H, int W, const bf16_t *__restrict__ d_col, float *__restrict__ d_rgb, size_t in_lo) {
    const float scale[3] = {0.458f, 0.448f, 0.450f};
    const size_t npix = (size_t)B * H * W;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
        const int x = (int)(p % W), y = (int)((p / W) % H);
        float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int qy = y - (tap / 3 - 1), qx = x - (tap % 3 - 1);
            if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
            const bf16_t *row = d_col + (p + (size_t)(qy - y) * W + (qx - x)) * 32 + 3 * tap;
#pragma unroll
            for (int c = 0; c < 3; c++) g[c] += load1(row + c, in_lo);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) d_rgb[3 * p + c] = g[c] * (2.f / scale[c]);
    }
}
// 1 x 1 convolution on NHWC rows: out[p][co] = sum_k in[p][k] w[co][k] (+ bias, ReLU), Cin a multiple of 32, Cout = 16 NT.
// wt [virtual chunk][Cout][32] bf16 (bf16x3: three virtual chunks per 32 input channels, (w_hi, w_hi, w_lo) for (x_hi, x_lo, x_hi), as the
// 3 x 3 kernels).  A wave owns 32 pixels (two MFMA tiles) x all output channels; operands go straight from global memory into the MFMA
// fragments (a lane reads the 16 bytes of k-group `kg` of its pixel / its output channel: the weights are a few KB and stay in L1).
template <bool RELU, int NT>
__global__ void __launch_bounds__(256) k_conv1x1_bf16(size_t npix, int Cin, const bf16_t *__restrict__ in, const bf16_t *__restrict__ wt, const float *__restrict__ bias,
                                                      bf16_t *__restrict__ out, size_t in_lo, size_t out_lo) {
    constexpr int Cout = 16 * NT;
    const int lane = threadIdx.x & 63, l15 = lane & 15, kg = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * 256) >> 6;
    const int nsrc = Cin / 32, nchunk = in_lo ? 3 * nsrc : nsrc;
    for (size_t p0 = wave * 32; p0 < npix; p0 += nwave * 32) {
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < nchunk; cc++) {
            const int sc = in_lo ? cc / 3 : cc;
            const bf16_t *plane = in + ((in_lo && cc % 3 == 1) ? in_lo : 0);
            bf16x8 bfrag[2], afrag[NT];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const size_t px = p0 + 16 * m + l15;
                bfrag[m] = px < npix ? *reinterpret_cast<const bf16x8 *>(plane + px * Cin + sc * 32 + kg * 8) : bf16x8{};
            }
#pragma unroll
            for (int n = 0; n < NT; n++) afrag[n] = *reinterpret_cast<const bf16x8 *>(wt + ((size_t)cc * Cout + tile_row_channel<NT>(n * 16 + l15)) * 32 + kg * 8);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < NT; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[n], bfrag[m], acc[m][n], 0, 0, 0);
        }
        // the lane holds the 4 NT consecutive channels 4 NT kg + 4 n + r of pixel l15 (tile_row_channel)
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const size_t px = p0 + 16 * m + l15;
            if (px >= npix) continue;
            const int co = 4 * NT * kg;
            float v[4 * NT];
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) v[4 * n + r] = acc[m][n][r];
            if (bias) {
#pragma unroll
                for (int k = 0; k < 4 * NT; k++) v[k] += bias[co + k];
            }
            if (RELU) {
#pragma unroll
                for (int k = 0; k < 4 * NT; k++) v[k] = fmaxf(v[k], 0.f);
            }
            store_row<NT>(out + px * Cout + co, out_lo, v);
        }
    }
}

// conv1_1 straight from the image: k_lpips_prepare_im2col + k_conv1x1_bf16<true, 4> in one kernel.  The im2col row of a pixel (3 x 3 taps x 3
// channels of ScalingLayer(2x - 1), zero-padded to 32) is built in registers by exactly the lanes that need it as a B fragment (pixel l15, channel
// group kg: the unit one thread of the prepare kernel writes), so the same values go through the same MFMA sequence -- bitwise the two-kernel
// result -- without 33 MB of rows written and read back per image and plane.  wt: [chunks][64][32], chunks = 1 (bf16) or 3 (bf16x3: hi, hi, lo).
#ifndef GOM_CONV1_1_STAGED
#define GOM_CONV1_1_STAGED 1   // 1: the wave's three scaled, split image rows are built ONCE in a wave-private LDS slab and the im2col fragments gathered from there
#endif                         // 0: every lane scales / splits its own 16 im2col values (round 4: 35 us per 512^2 image, half of it that arithmetic, nine-fold redundant)
template <bool X3>
__global__ void __launch_bounds__(256) k_conv1_1_image(int B, int H, int W, const float *__restrict__ rgb, const bf16_t *__restrict__ wt, const float *__restrict__ bias,
                                                       bf16_t *__restrict__ out, size_t out_lo) {
    constexpr int NT = 4, Cout = 64;
    const float shift[3] = {-0.030f, -0.088f, -0.188f}, scale[3] = {0.458f, 0.448f, 0.450f};
    const int lane = threadIdx.x & 63, l15 = lane & 15, kg = lane >> 4;
#if GOM_CONV1_1_STAGED
    // S[ky][i]: channel i % 3 of pixel x0 - 1 + i / 3 of image row y + ky - 1, scaled, as (hi | lo << 16); zero outside the image.  The im2col row of pixel x0 + xl
    // is S[0][3 xl .. 3 xl + 8], S[1][..], S[2][..]: column k = 9 ky + j of it is S[ky][3 xl + j]
    __shared__ uint32_t s_rows[4][3][104];
    uint32_t (*S)[104] = s_rows[threadIdx.x >> 6];
#endif
    // a wave owns 32 consecutive pixels of ONE image row (W is a multiple of 32 here: the launcher checks): 32-bit index arithmetic, once per wave
    const uint32_t wpr = (uint32_t)W / 32u, nrow = (uint32_t)B * (uint32_t)H;
    const uint32_t wave0 = (blockIdx.x * 256u + threadIdx.x) >> 6, nwave = (gridDim.x * 256u) >> 6;
    for (uint32_t wv = wave0; wv < nrow * wpr; wv += nwave) {
        const uint32_t row = wv / wpr, x0 = (wv - row * wpr) * 32u;
        const int y = (int)(row % (uint32_t)H);
        const size_t p0 = (size_t)row * W + x0;
        const float *img_row = rgb + 3 * ((size_t)row * W);   // pixel (row, 0) of this image row; the rows above / below are +- 3 W floats
        bf16x8 bhi[2], blo[2];
#if GOM_CONV1_1_STAGED
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const int dy = ky - 1;
            const bool row_in = y + dy >= 0 && y + dy < H;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = lane + 64 * h;
                if (i < 102) {
                    const int px = i / 3, c = i - 3 * px, xx = (int)x0 - 1 + px;
                    float v = 0.f;
                    if (row_in && xx >= 0 && xx < W) {
                        const float sh = c == 0 ? shift[0] : (c == 1 ? shift[1] : shift[2]), sc = c == 0 ? scale[0] : (c == 1 ? scale[1] : scale[2]);
                        v = ((2.f * img_row[3 * ((ptrdiff_t)dy * W + xx) + c] - 1.f) - sh) / sc;
                    }
                    bf16_t hh, ll = 0;
                    if (X3) split_bf(v, hh, ll); else hh = f2bf(v);
                    S[ky][i] = (uint32_t)hh | ((uint32_t)ll << 16);
                }
            }
        }
        // (LDS operations of one wave execute in order: no barrier)
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int xl = 16 * m + l15;
            uint32_t w8[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = kg * 8 + r, ky = (k >= 9 ? 1 : 0) + (k >= 18 ? 1 : 0), j = k - 9 * ky;
                w8[r] = k < 27 ? S[ky][3 * xl + j] : 0u;
            }
            const uint4 qh = make_uint4((w8[0] & 0xffffu) | (w8[1] << 16), (w8[2] & 0xffffu) | (w8[3] << 16), (w8[4] & 0xffffu) | (w8[5] << 16), (w8[6] & 0xffffu) | (w8[7] << 16));
            const uint4 ql = make_uint4((w8[0] >> 16) | (w8[1] & 0xffff0000u), (w8[2] >> 16) | (w8[3] & 0xffff0000u), (w8[4] >> 16) | (w8[5] & 0xffff0000u),
                                        (w8[6] >> 16) | (w8[7] & 0xffff0000u));
            bhi[m] = __builtin_bit_cast(bf16x8, qh);
            blo[m] = __builtin_bit_cast(bf16x8, ql);
        }
#else
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int x = (int)x0 + 16 * m + l15;
            bf16_t hi[8], lo[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int k = kg * 8 + r;
                float v = 0.f;
                if (k < 27) {
                    const int tap = k / 3, c = k % 3;
                    const int dy = tap / 3 - 1, xx = x + tap % 3 - 1;
                    if (y + dy >= 0 && y + dy < H && xx >= 0 && xx < W) v = ((2.f * img_row[3 * ((ptrdiff_t)dy * W + xx) + c] - 1.f) - shift[c]) / scale[c];
                }
                if (X3) split_bf(v, hi[r], lo[r]); else { hi[r] = f2bf(v); lo[r] = 0; }
            }
            const uint4 qh = make_uint4((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16), (uint32_t)hi[4] | ((uint32_t)hi[5] << 16),
                                        (uint32_t)hi[6] | ((uint32_t)hi[7] << 16));
            const uint4 ql = make_uint4((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16), (uint32_t)lo[4] | ((uint32_t)lo[5] << 16),
                                        (uint32_t)lo[6] | ((uint32_t)lo[7] << 16));
            bhi[m] = __builtin_bit_cast(bf16x8, qh);
            blo[m] = __builtin_bit_cast(bf16x8, ql);
        }
#endif
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < (X3 ? 3 : 1); cc++) {   // (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo): the order of k_conv1x1_bf16's chunk loop
            bf16x8 afrag[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) afrag[n] = *reinterpret_cast<const bf16x8 *>(wt + ((size_t)cc * Cout + tile_row_channel<NT>(n * 16 + l15)) * 32 + kg * 8);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < NT; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[n], cc == 1 ? blo[m] : bhi[m], acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const size_t px = p0 + 16 * m + l15;
            const int co = 16 * kg;   // 16 consecutive channels (tile_row_channel<4>)
            float v[16];
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) v[4 * n + r] = fmaxf(acc[m][n][r] + bias[co + 4 * n + r], 0.f);
            store_row<4>(out + px * Cout + co, out_lo, v);
        }
    }
}

// conv1_1's backward-data pass straight to the image gradient (round 5): k_conv1x1_bf16<false, 2> (d(im2col rows) = W^T dY, 64 -> 32 columns) and
// k_lpips_unprepare_col2im in ONE kernel.  A workgroup owns a 16 x 16 pixel tile: the im2col gradient rows of its 18 x 18 halo pixels come off the matrix
// cores (the fragments and the MFMA order of the 1 x 1 kernel: (g_hi, w_hi), (g_lo, w_hi), (g_hi, w_lo) per 32 channels) and stay in LDS as fp32 -- the
// two-kernel form rounded them to two bf16 planes in between -- and every pixel gathers column 3 t + c from the neighbour that has it as tap t, in the
// col2im kernel's order.  70 us of two HBM streams (67 MB read, 34 MB written and read back) become one pass over the 64-channel gradient.
template <bool X3>
__global__ void __launch_bounds__(256) k_conv1_1_bwd_image(int B, int H, int W, const bf16_t *__restrict__ g, const bf16_t *__restrict__ wt, float *__restrict__ d_rgb,
                                                           size_t in_lo) {
    constexpr int NT = 2, Cout = 32, Cin = 64, PW = 18, NP = PW * PW, STR = 33;   // (33 words per halo pixel: the gather's 16 consecutive pixels hit 16 banks)
    __shared__ float s_col[NP * STR];
    const float scale[3] = {0.458f, 0.448f, 0.450f};
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    const int tx = (int)(blockIdx.x % (uint32_t)tiles_x), ty = (int)((blockIdx.x / (uint32_t)tiles_x) % (uint32_t)tiles_y), b = (int)(blockIdx.x / (uint32_t)(tiles_x * tiles_y));
    const int lane = threadIdx.x & 63, l15 = lane & 15, kg = lane >> 4, wave = threadIdx.x >> 6;
    const int x0 = tx * 16 - 1, y0 = ty * 16 - 1;
    constexpr int nsrc = Cin / 32, nchunk = X3 ? 3 * nsrc : nsrc;
    bf16x8 afrag[nchunk][NT];   // the weights: a few KB, the same for every tile
#pragma unroll
    for (int cc = 0; cc < nchunk; cc++)
#pragma unroll
        for (int n = 0; n < NT; n++) afrag[cc][n] = *reinterpret_cast<const bf16x8 *>(wt + ((size_t)cc * Cout + tile_row_channel<NT>(n * 16 + l15)) * 32 + kg * 8);
    for (int t = wave; t < (NP + 15) / 16; t += 4) {
        const int j = t * 16 + l15, ly = j / PW, lx = j - ly * PW, gy = y0 + ly, gx = x0 + lx;
        const bool in = j < NP && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const bf16_t *src = g + (((size_t)b * H + (in ? gy : 0)) * W + (in ? gx : 0)) * Cin + kg * 8;
        bf16x8 hi[nsrc], lo[nsrc];
#pragma unroll
        for (int sc = 0; sc < nsrc; sc++) {
            hi[sc] = in ? *reinterpret_cast<const bf16x8 *>(src + sc * 32) : bf16x8{};
            lo[sc] = (X3 && in) ? *reinterpret_cast<const bf16x8 *>(src + in_lo + sc * 32) : bf16x8{};
        }
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < nchunk; cc++) {
            const int sc = X3 ? cc / 3 : cc;
            const bf16x8 bfrag = (X3 && cc % 3 == 1) ? lo[sc] : hi[sc];
#pragma unroll
            for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[cc][n], bfrag, acc[n], 0, 0, 0);
        }
        if (j < NP) {   // the lane holds columns 8 kg + 4 n + r of halo pixel j (tile_row_channel<2>)
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) s_col[j * STR + 8 * kg + 4 * n + r] = acc[n][r];
        }
    }
    __syncthreads();
    const int px = threadIdx.x & 15, py = threadIdx.x >> 4, gx = tx * 16 + px, gy = ty * 16 + py;
    float gs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {   // pixel q = p - offset(tap) has p as its tap `tap` (an outside q left zeros: its fragments were zero)
        const float *row = s_col + ((py + 2 - tap / 3) * PW + (px + 2 - tap % 3)) * STR + 3 * tap;
#pragma unroll
        for (int c = 0; c < 3; c++) gs[c] += row[c];
    }
    if (gx < W && gy < H) {
        float *dst = d_rgb + 3 * (((size_t)b * H + gy) * W + gx);
#pragma unroll
        for (int c = 0; c < 3; c++) dst[c] = gs[c] * (2.f / scale[c]);
    }
}

// ---- LPIPS head on NHWC bf16 taps: 16 lanes per pixel, each lane strides over the channels 8 at a time ----------------
constexpr float kEps = 1e-10f;
__device__ __forceinline__ float sum16(float v) {  // over the 16 lanes of a DPP row
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
template <bool BWD>
__global__ void __launch_bounds__(256) k_lpips_head_nhwc(int C, size_t HW, const bf16_t *__restrict__ f0, const bf16_t *__restrict__ f1,
                                                         const float *__restrict__ w, const float *__restrict__ grad_out,
                                                         float *__restrict__ partials, bf16_t *__restrict__ d_f0) {
    __shared__ float s_red[4];
    const size_t b = blockIdx.y;
    f0 += b * HW * C; f1 += b * HW * C;
    if (BWD) d_f0 += b * HW * C;
    const int sub = threadIdx.x & 15;                 // lane inside the pixel's 16-lane group
    const size_t grp = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4, ngrp = ((size_t)gridDim.x * 256) >> 4;
    const float go = BWD ? grad_out[b] * (2.0f / (float)HW) : 0.f;
    float acc = 0.f;
    for (size_t p = grp; p < HW; p += ngrp) {
        const bf16_t *a = f0 + p * C, *bb = f1 + p * C;
        float s0 = 0.f, s1 = 0.f;
        for (int c = sub * 8; c < C; c += 128) {
            float x[8], y[8];
            unpack8(*reinterpret_cast<const uint4 *>(a + c), x);
            unpack8(*reinterpret_cast<const uint4 *>(bb + c), y);
#pragma unroll
            for (int k = 0; k < 8; k++) { s0 += x[k] * x[k]; s1 += y[k] * y[k]; }
        }
        s0 = sum16(s0); s1 = sum16(s1);
        const float n0 = sqrtf(s0 + kEps);
        const float i0 = 1.f / (n0 + kEps), i1 = 1.f / (sqrtf(s1 + kEps) + kEps);
        float v = 0.f;   // forward: sum w d^2 ; backward: sum w d f0
        for (int c = sub * 8; c < C; c += 128) {
            float x[8], y[8];
            unpack8(*reinterpret_cast<const uint4 *>(a + c), x);
            unpack8(*reinterpret_cast<const uint4 *>(bb + c), y);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float d = x[k] * i0 - y[k] * i1;
                v += BWD ? w[c + k] * d * x[k] : w[c + k] * d * d;
            }
        }
        v = sum16(v);
        if (!BWD) {
            acc += (sub == 0) ? v : 0.f;
        } else {
            const float kk = v * i0 * i0 / n0;
            for (int c = sub * 8; c < C; c += 128) {
                float x[8], y[8];
                unpack8(*reinterpret_cast<const uint4 *>(a + c), x);
                unpack8(*reinterpret_cast<const uint4 *>(bb + c), y);
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // times the ReLU derivative of the tap's own layer (x = 0 <=> the pre-activation was clipped)
                    const float g0 = x[2 * k] > 0.f ? go * (w[c + 2 * k] * (x[2 * k] * i0 - y[2 * k] * i1) * i0 - x[2 * k] * kk) : 0.f;
                    const float g1 = x[2 * k + 1] > 0.f ? go * (w[c + 2 * k + 1] * (x[2 * k + 1] * i0 - y[2 * k + 1] * i1) * i0 - x[2 * k + 1] * kk) : 0.f;
                    o[k] = (uint32_t)f2bf(g0) | ((uint32_t)f2bf(g1) << 16);
                }
                *reinterpret_cast<uint4 *>(d_f0 + p * C + c) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)HW;
    }
}

// Single-pass variant for the tap widths of VGG16 (C = 64 .. 512): LPP = C / 8 lanes per pixel, every lane keeps its 8 channels
// of both feature maps in registers across the three phases (norms, weighted difference, gradient), so each feature is read
// from memory once (the generic kernel above reads it two / three times and idles half its lanes at C = 64).
// VAL (with BWD): the backward pass also leaves the tap's VALUE (per-block sums in `partials`, one per workgroup of ITS grid): a training step reads the
// two feature maps once instead of twice (gom_lpips_layer_backward_value_planes; the forward launch is skipped).
// POOL (with BWD): f0 is also the input of a 2 x 2 max-pool whose output gradient `dy` [HW / 4][C] exists already (the layers above were walked first):
// the pixel that is the FIRST maximum of its window (row-major order, like the reference framework) and positive receives it here, added in fp32 before
// the one rounding to the gradient's planes -- k_maxpool2_bwd's accumulate form without its read-modify-write of the whole gradient tensor.
template <bool BWD, int LPP, bool VAL = false, bool POOL = false>
__global__ void __launch_bounds__(BWD ? 256 : 1024) k_lpips_head_nhwc_1p(size_t HW, const bf16_t *__restrict__ f0, const bf16_t *__restrict__ f1,
                                                                        const float *__restrict__ w, const float *__restrict__ grad_out,
                                                                        float *__restrict__ partials, bf16_t *__restrict__ d_f0, size_t f_lo, size_t d_lo,
                                                                        const bf16_t *__restrict__ dy = nullptr, size_t dy_lo = 0, int W = 0) {
    constexpr int C = 8 * LPP;
    __shared__ float s_red[16];
    const size_t b = blockIdx.y;
    f0 += b * HW * C; f1 += b * HW * C;
    if (BWD) d_f0 += b * HW * C;
    if (POOL) dy += b * (HW / 4) * C;
    const int sub = threadIdx.x & (LPP - 1);
    const size_t grp = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LPP, ngrp = ((size_t)gridDim.x * blockDim.x) / LPP;
    const float go = BWD ? grad_out[b] * (2.0f / (float)HW) : 0.f;
    float wl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) wl[k] = w[sub * 8 + k];
    float acc = 0.f;
    // one pixel: x = its 8 channels of f0 (already loaded), add = what else its gradient receives (the pool's routing, or zeros)
    auto pixel = [&](size_t p, const float (&x)[8], const float (&add)[8]) {
        float y[8];
        load8(f1 + p * C + sub * 8, f_lo, y);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) { s0 += x[k] * x[k]; s1 += y[k] * y[k]; }
#pragma unroll
        for (int d = LPP / 2; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); }
        const float n0 = sqrtf(s0 + kEps);
        const float i0 = 1.f / (n0 + kEps), i1 = 1.f / (sqrtf(s1 + kEps) + kEps);
        float v = 0.f, v2 = 0.f;   // forward: sum w d^2 ; backward: sum w d f0 (v2: the forward's sum beside it)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float d = x[k] * i0 - y[k] * i1;
            v += BWD ? wl[k] * d * x[k] : wl[k] * d * d;
            if (BWD && VAL) v2 += wl[k] * d * d;
        }
#pragma unroll
        for (int d = LPP / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (BWD && VAL) {
#pragma unroll
            for (int d = LPP / 2; d >= 1; d >>= 1) v2 += __shfl_xor(v2, d, 64);
            acc += (sub == 0) ? v2 : 0.f;
        }
        if (!BWD) {
            acc += (sub == 0) ? v : 0.f;
        } else {
            const float kk = v * i0 * i0 / n0;
            float gq[8];
#pragma unroll
            for (int k = 0; k < 8; k++)   // times the ReLU derivative of the tap's own layer (x = 0 <=> the pre-activation was clipped)
                gq[k] = (x[k] > 0.f ? go * (wl[k] * (x[k] * i0 - y[k] * i1) * i0 - x[k] * kk) : 0.f) + add[k];
            store8(d_f0 + p * C + sub * 8, d_lo, gq);
        }
    };
    if constexpr (POOL) {
        // a group of LPP lanes owns a 2 x 2 WINDOW: its four pixels' f0 rows are read once, the window's maximum is found per channel, then the four
        // pixels are walked like everywhere else
        const size_t NW = HW / 4;
        const int W2 = W >> 1;
        for (size_t wi = grp; wi < NW; wi += ngrp) {
            const size_t wy = wi / (size_t)W2, wx = wi - wy * W2;
            const size_t p0 = 2 * wy * W + 2 * wx;
            const size_t pq[4] = {p0, p0 + 1, p0 + W, p0 + W + 1};
            float xs[4][8], g8[8];
#pragma unroll
            for (int q = 0; q < 4; q++) load8(f0 + pq[q] * C + sub * 8, f_lo, xs[q]);
            load8(dy + wi * C + sub * 8, dy_lo, g8);
            int am[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {   // first maximum in row-major order; nothing passes a non-positive one (the ReLU derivative)
                float best = xs[0][k];
                am[k] = 0;
#pragma unroll
                for (int q = 1; q < 4; q++)
                    if (xs[q][k] > best) { best = xs[q][k]; am[k] = q; }
                if (!(best > 0.f)) am[k] = -1;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float add[8];
#pragma unroll
                for (int k = 0; k < 8; k++) add[k] = am[k] == q ? g8[k] : 0.f;
                pixel(pq[q], xs[q], add);
            }
        }
    } else {
        const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (size_t p = grp; p < HW; p += ngrp) {
            float x[8];
            load8(f0 + p * C + sub * 8, f_lo, x);
            pixel(p, x, zero);
        }
    }
    if (!BWD || VAL) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += s_red[k];
            partials[b * (VAL ? (size_t)GOM_LPIPS_HEAD_BLOCKS : (size_t)gridDim.x) + blockIdx.x] = t / (float)HW;
        }
    }
}

// values of the taps whose per-block sums came out of the backward kernels (k_lpips_head_nhwc_1p<true, .., true>): one workgroup per (tap, image) adds
// its nblk[tap] sums in a fixed order and leaves the total in slot 0 of the caller's [5][B][GOM_LOSS_BLOCKS] rows (the other slots: zero)
__global__ void __launch_bounds__(256) k_lpips_fold_values(int B, const float *__restrict__ scratch, int scratch_stride, int n0, int n1, int n2, int n3, int n4,
                                                           float *__restrict__ partials) {
    __shared__ float s_red[256];
    const int t = blockIdx.x, b = blockIdx.y;
    const int n = t == 0 ? n0 : (t == 1 ? n1 : (t == 2 ? n2 : (t == 3 ? n3 : n4)));
    const float *src = scratch + ((size_t)t * B + b) * scratch_stride;
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += src[i];
    s_red[threadIdx.x] = a;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
        __syncthreads();
    }
    float *dst = partials + ((size_t)t * B + b) * GOM_LOSS_BLOCKS;
    for (int i = threadIdx.x; i < GOM_LOSS_BLOCKS; i += 256) dst[i] = i == 0 ? s_red[0] : 0.f;
}

}  // namespace

extern "C" int gom_conv3x3_bf16(int B, int H, int W, int Cin, int Cout, const void *in, const void *wt, const float *bias, const void *mask,
                                void *out, uint32_t flags, void *stream) {
    return gom_conv3x3_bf16_splitk(B, H, W, Cin, Cout, in, wt, bias, mask, out, flags, 1, nullptr, stream);
}

extern "C" int gom_conv3x3_bf16_splitk(int B, int H, int W, int Cin, int Cout, const void *in, const void *wt, const float *bias, const void *mask,
                                       void *out, uint32_t flags, int splits, float *workspace, void *stream) {
    return gom_conv3x3_planes(B, H, W, Cin, Cout, in, wt, bias, mask, out, flags, splits, workspace, 0, 0, nullptr, 0, stream);
}

extern "C" int gom_conv3x3_splits(int B, int H, int W, int Cin, int Cout) {
    // 16 x 16-pixel tiles x 64 output channels: split the input channels until the launch has >= 256 workgroups (one per CU; equal shares of the
    // chunks).  Every split layer pays a second launch (k_splitk_epilogue, 8-14 us).  Measured on the Model iteration (MI355X, bf16x3, target trunk
    // prefetched = one image per pass): a target of 256 workgroups 2.84 ms, 384 / 512 (two per CU) 2.95-2.98, the round-3 rule (512 on 8-row tiles) 2.87-2.92;
    // below 256 the launcher falls back to 8-row tiles through the unpipelined kernel: 4.4 ms.
    static const int mode = getenv("GOM_CONV_SPLIT_MODE") ? atoi(getenv("GOM_CONV_SPLIT_MODE")) : 2;   // development switch: 0 = round-3 rule, 1 = 512 on 16-row tiles
    const long blocks = (long)((W + kTileW - 1) / kTileW) * (mode ? (H + 15) / 16 : (H + 7) / 8) * (Cout / kBN) * B;
    const long want = mode == 2 ? 256 : 512;
    int s = 1;
    while (s < 16 && blocks * s < want && (Cin / kKC) % (2 * s) == 0) s *= 2;
    return s;
}

// in_lo / out_lo: element offsets of the lo planes of input and output (0: plain bf16); bf16x3 takes `wt` with 3 x Cin/32 chunks
// (w_hi, w_hi, w_lo per 32 input channels: lpips.pack_conv_weight_x3)
int gom_conv3x3_planes(int B, int H, int W, int Cin, int Cout, const void *in, const void *wt, const float *bias, const void *mask,
                       void *out, uint32_t flags, int splits, float *workspace, size_t in_lo, size_t out_lo, void *pooled, size_t pooled_lo, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kKC || Cout % kBN) { gom_set_error("gom_conv3x3_bf16: Cin %% 32 / Cout %% 64 / sizes"); return -1; }
    if (!in || !wt || !out) { gom_set_error("gom_conv3x3_bf16: null pointer"); return -1; }
    if (splits < 1 || (splits > 1 && (!workspace || (Cin / kKC) % splits))) { gom_set_error("gom_conv3x3_bf16: bad split-K arguments"); return -1; }
    if (pooled && ((H & 1) || (W & 1) || mask)) { gom_set_error("gom_conv3x3_bf16: fused 2 x 2 pooling needs even H, W and no mask"); return -1; }
    bf16_t *p_ = (bf16_t *)pooled;
    hipStream_t st = (hipStream_t)stream;
    // 16 pixel rows per workgroup when that still leaves >= 2 workgroups per CU
    const long blocks16 = (long)((W + kTileW - 1) / kTileW) * ((H + 15) / 16) * (Cout / kBN) * B * splits;
    const int TH = (blocks16 >= 256 && H >= 16) ? 16 : 8;
    const dim3 grid(((W + kTileW - 1) / kTileW) * ((H + TH - 1) / TH), Cout / kBN, B * splits);
    const bf16_t *i_ = (const bf16_t *)in, *w_ = (const bf16_t *)wt, *m_ = (const bf16_t *)mask;
    static const bool use_v2 = !(getenv("GOM_CONV_V2") && atoi(getenv("GOM_CONV_V2")) == 0);   // development switches
    static const int v2_rpw = getenv("GOM_CONV_RPW") ? atoi(getenv("GOM_CONV_RPW")) : 2;
    // bf16x3 with shared stages (k_conv3x3_x3s: one workgroup per CU) for launches of FEWER than two workgroups per CU -- the deep layers and every split-K
    // launch: conv4 forward 130 -> 102 us, conv5 41 -> 34, conv3 / conv4 backward 72 -> 57; launches of >= 512 workgroups keep k_conv3x3_bf16_v2, whose second
    // co-resident workgroup hides the prologue and the store tail of the first (conv1_2: 130 us against 161, conv2_2 112 against 131: LABBOOK R5.3).
    // GOM_CONV_X3S: 0 = never, 2 = always (development)
    static const int x3s_mode = getenv("GOM_CONV_X3S") ? atoi(getenv("GOM_CONV_X3S")) : 1;
    // (Measured and dropped: RPW = 4 -- four waves of four rows, 0.33 fragment reads per MFMA, 232 registers, one wave per SIMD: Model iteration 2.85 against
    //  2.78 ms; the same products with ONE-TAP weight stages and single-buffered patches in 66 KB, two workgroups per CU, for the launches of >= 512
    //  workgroups (k_conv3x3_x3l): 2.93-2.96 against 2.93-3.05 ms on the same box -- no gain over k_conv3x3_bf16_v2 there: LABBOOK R5.3.)
    const bool use_x3s = x3s_mode == 2 || (x3s_mode == 1 && (long)grid.x * grid.y * grid.z < 512);
#define GOM_CONV_LAUNCH(RELU_, SPLIT_, ...)                                                                                          \
    do {                                                                                                                              \
        if (TH == 16 && use_v2 && use_x3s && in_lo) {                                                                                 \
            static const hipError_t attrx_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_x3s<RELU_, SPLIT_, 2>),    \
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kX3Lds);                  \
            if (attrx_ != hipSuccess) { gom_set_error("hipFuncSetAttribute(k_conv3x3_x3s) failed"); return -1; }                      \
            hipLaunchKernelGGL((k_conv3x3_x3s<RELU_, SPLIT_, 2>), grid, dim3(512), kX3Lds, st, __VA_ARGS__);                          \
        } else if (TH == 16 && use_v2) {                                                                                                     \
            static const hipError_t attr2_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_bf16_v2<RELU_, SPLIT_, 2>),   \
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kV2Lds);                  \
            static const hipError_t attr4_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_bf16_v2<RELU_, SPLIT_, 4>),   \
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kV2Lds);                  \
            if (attr2_ != hipSuccess || attr4_ != hipSuccess) { gom_set_error("hipFuncSetAttribute(k_conv3x3_bf16_v2) failed"); return -1; } \
            if (v2_rpw == 4) hipLaunchKernelGGL((k_conv3x3_bf16_v2<RELU_, SPLIT_, 4>), grid, dim3(256), kV2Lds, st, __VA_ARGS__);     \
            else hipLaunchKernelGGL((k_conv3x3_bf16_v2<RELU_, SPLIT_, 2>), grid, dim3(512), kV2Lds, st, __VA_ARGS__);                 \
        } else if (TH == 16) hipLaunchKernelGGL((k_conv3x3_bf16<RELU_, SPLIT_, 16>), grid, dim3(512), 0, st, __VA_ARGS__);           \
        else hipLaunchKernelGGL((k_conv3x3_bf16<RELU_, SPLIT_, 8>), grid, dim3(256), 0, st, __VA_ARGS__);                            \
    } while (0)
    if (splits > 1) {
        GOM_CONV_LAUNCH(false, true, H, W, Cin, Cout, i_, w_, nullptr, nullptr, nullptr, splits, workspace, in_lo, out_lo, nullptr, (size_t)0);
        GOM_LAUNCH_CHECK();
        const size_t n8 = (size_t)B * H * W * Cout / 8;
        if (p_) hipLaunchKernelGGL(k_splitk_epilogue<true>, dim3((unsigned)((n8 / 4 + 255) / 256 < 4096 ? (n8 / 4 + 255) / 256 : 4096)), dim3(256), 0, st, n8, H, W, Cout, splits,
                                   workspace, bias, m_, (bf16_t *)out, (flags & GOM_CONV_RELU) ? 1 : 0, out_lo, p_, pooled_lo);
        else hipLaunchKernelGGL(k_splitk_epilogue<false>, dim3((unsigned)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096)), dim3(256), 0, st, n8, H, W, Cout, splits,
                                workspace, bias, m_, (bf16_t *)out, (flags & GOM_CONV_RELU) ? 1 : 0, out_lo, p_, pooled_lo);
    } else if (flags & GOM_CONV_RELU) {
        GOM_CONV_LAUNCH(true, false, H, W, Cin, Cout, i_, w_, bias, m_, (bf16_t *)out, 1, nullptr, in_lo, out_lo, p_, pooled_lo);
    } else {
        GOM_CONV_LAUNCH(false, false, H, W, Cin, Cout, i_, w_, bias, m_, (bf16_t *)out, 1, nullptr, in_lo, out_lo, p_, pooled_lo);
    }
#undef GOM_CONV_LAUNCH
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_maxpool2x2_bf16(int B, int H, int W, int C, const void *x, void *y, void *stream) { return gom_maxpool2x2_planes(B, H, W, C, x, y, 0, 0, stream); }
int gom_maxpool2x2_planes(int B, int H, int W, int C, const void *x, void *y, size_t x_lo, size_t y_lo, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C % 8) { gom_set_error("gom_maxpool2x2_bf16: bad sizes"); return -1; }
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(k_maxpool2_fwd, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream, B, H, W, C,
                       (const bf16_t *)x, (bf16_t *)y, x_lo, y_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_maxpool2x2_backward_bf16(int B, int H, int W, int C, const void *x, const void *dy, void *dx, int accumulate, void *stream) {
    return gom_maxpool2x2_backward_planes(B, H, W, C, x, dy, dx, accumulate, 0, 0, 0, stream);
}
int gom_maxpool2x2_backward_planes(int B, int H, int W, int C, const void *x, const void *dy, void *dx, int accumulate, size_t x_lo, size_t dy_lo, size_t dx_lo, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) { gom_set_error("gom_maxpool2x2_backward_bf16: bad sizes (even H, W; C a multiple of 8)"); return -1; }
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(k_maxpool2_bwd, dim3((unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0, (hipStream_t)stream, B, H, W, C,
                       (const bf16_t *)x, (const bf16_t *)dy, (bf16_t *)dx, accumulate, x_lo, dy_lo, dx_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_lpips_prepare_bf16(int B, int H, int W, const float *rgb, void *out32, void *stream) { return gom_lpips_prepare_planes(B, H, W, rgb, out32, 0, stream); }
int gom_lpips_prepare_planes(int B, int H, int W, const float *rgb, void *out32, size_t out_lo, void *stream) {
    const size_t npix = (size_t)B * H * W;
    if (!rgb || !out32 || npix == 0) { gom_set_error("gom_lpips_prepare_bf16: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_lpips_prepare, dim3((unsigned)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, npix, rgb,
                       (bf16_t *)out32, out_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}

// first layer as im2col rows + 1 x 1 convolutions (see k_lpips_prepare_im2col)
int gom_lpips_prepare_im2col_planes(int B, int H, int W, const float *rgb, void *out32, size_t out_lo, void *stream) {
    const size_t units = (size_t)B * H * W * 4;
    if (!rgb || !out32 || units == 0) { gom_set_error("gom_lpips_prepare_im2col: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_lpips_prepare_im2col, dim3((unsigned)((units + 255) / 256 < 8192 ? (units + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream, B, H, W, rgb,
                       (bf16_t *)out32, out_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}
int gom_lpips_unprepare_col2im_planes(int B, int H, int W, const void *d_col, float *d_rgb, size_t in_lo, void *stream) {
    const size_t npix = (size_t)B * H * W;
    if (!d_col || !d_rgb || npix == 0) { gom_set_error("gom_lpips_unprepare_col2im: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_lpips_unprepare_col2im, dim3((unsigned)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, B, H, W,
                       (const bf16_t *)d_col, d_rgb, in_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}
int gom_conv1x1_planes(size_t npix, int Cin, int Cout, const void *in, const void *wt, const float *bias, void *out, int relu, size_t in_lo, size_t out_lo, void *stream) {
    if (!in || !wt || !out || npix == 0 || Cin % 32 || (Cout != 32 && Cout != 64)) { gom_set_error("gom_conv1x1: Cin a multiple of 32, Cout 32 or 64"); return -1; }
    const size_t waves = (npix + 31) / 32;
    const unsigned grid = (unsigned)((waves + 3) / 4 < 8192 ? (waves + 3) / 4 : 8192);
#define GOM_C1(RL, NT_) hipLaunchKernelGGL((k_conv1x1_bf16<RL, NT_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, npix, Cin, (const bf16_t *)in, (const bf16_t *)wt, bias, (bf16_t *)out, in_lo, out_lo)
    if (Cout == 64) { if (relu) GOM_C1(true, 4); else GOM_C1(false, 4); }
    else { if (relu) GOM_C1(true, 2); else GOM_C1(false, 2); }
#undef GOM_C1
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_conv1_1_bwd_image_planes(int B, int H, int W, const void *g, const void *wt, float *d_rgb, size_t in_lo, void *stream) {
    if (!g || !wt || !d_rgb || B <= 0 || H <= 0 || W <= 0) { gom_set_error("gom_conv1_1_bwd_image: bad arguments"); return -1; }
    const unsigned grid = (unsigned)B * (unsigned)((H + 15) / 16) * (unsigned)((W + 15) / 16);
    if (in_lo) hipLaunchKernelGGL((k_conv1_1_bwd_image<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, B, H, W, (const bf16_t *)g, (const bf16_t *)wt, d_rgb, in_lo);
    else hipLaunchKernelGGL((k_conv1_1_bwd_image<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, B, H, W, (const bf16_t *)g, (const bf16_t *)wt, d_rgb, in_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_conv1_1_image_planes(int B, int H, int W, const float *rgb, const void *wt, const float *bias, void *out, size_t out_lo, void *stream) {
    const size_t npix = (size_t)B * H * W;
    if (!rgb || !wt || !bias || !out || npix == 0 || W % 32) { gom_set_error("gom_conv1_1_image: bad arguments (W must be a multiple of 32)"); return -1; }
    const size_t waves = (npix + 31) / 32;
    const unsigned grid = (unsigned)((waves + 3) / 4 < 8192 ? (waves + 3) / 4 : 8192);
    if (out_lo) hipLaunchKernelGGL((k_conv1_1_image<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, B, H, W, rgb, (const bf16_t *)wt, bias, (bf16_t *)out, out_lo);
    else hipLaunchKernelGGL((k_conv1_1_image<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, B, H, W, rgb, (const bf16_t *)wt, bias, (bf16_t *)out, out_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_lpips_unprepare_bf16(int B, int H, int W, int Cpad, const void *d_in, float *d_rgb, void *stream) { return gom_lpips_unprepare_planes(B, H, W, Cpad, d_in, d_rgb, 0, stream); }
int gom_lpips_unprepare_planes(int B, int H, int W, int Cpad, const void *d_in, float *d_rgb, size_t in_lo, void *stream) {
    const size_t npix = (size_t)B * H * W;
    if (!d_in || !d_rgb || npix == 0 || Cpad < 3) { gom_set_error("gom_lpips_unprepare_bf16: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_lpips_unprepare, dim3((unsigned)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, npix, Cpad,
                       (const bf16_t *)d_in, d_rgb, in_lo);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_lpips_layer_forward_nhwc_bf16(int B, int C, int HW, const void *f0, const void *f1, const float *w, float *partials, void *stream) {
    return gom_lpips_layer_forward_planes(B, C, HW, f0, f1, w, partials, 0, stream);
}
int gom_lpips_layer_forward_planes(int B, int C, int HW, const void *f0, const void *f1, const float *w, float *partials, size_t f_lo, void *stream) {
    if (f_lo && C != 64 && C != 128 && C != 256 && C != 512) { gom_set_error("bf16x3 LPIPS head: C must be 64, 128, 256 or 512"); return -1; }
    if (B <= 0 || C <= 0 || (C % 128 && C != 64) || HW <= 0) { gom_set_error("gom_lpips_layer_forward_nhwc_bf16: C must be 64 or a multiple of 128"); return -1; }
#define GOM_HEAD1P(BWD_, LPP_, GRID_, NT_, ...) hipLaunchKernelGGL((k_lpips_head_nhwc_1p<BWD_, LPP_>), GRID_, dim3(NT_), 0, (hipStream_t)stream, __VA_ARGS__)
    const dim3 grid(GOM_LOSS_BLOCKS, B);
    if (C == 64) GOM_HEAD1P(false, 8, grid, 1024, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, nullptr, partials, nullptr, f_lo, (size_t)0);
    else if (C == 128) GOM_HEAD1P(false, 16, grid, 1024, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, nullptr, partials, nullptr, f_lo, (size_t)0);
    else if (C == 256) GOM_HEAD1P(false, 32, grid, 1024, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, nullptr, partials, nullptr, f_lo, (size_t)0);
    else if (C == 512) GOM_HEAD1P(false, 64, grid, 1024, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, nullptr, partials, nullptr, f_lo, (size_t)0);
    else
    hipLaunchKernelGGL(k_lpips_head_nhwc<false>, dim3(GOM_LOSS_BLOCKS, B), dim3(256), 0, (hipStream_t)stream, C, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w,
                       nullptr, partials, nullptr);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_lpips_layer_backward_nhwc_bf16(int B, int C, int HW, const void *f0, const void *f1, const float *w, const float *grad_out, void *d_f0,
                                                  void *stream) {
    return gom_lpips_layer_backward_planes(B, C, HW, f0, f1, w, grad_out, d_f0, 0, 0, stream);
}
// backward of a tap that also leaves the tap's value: `block_sums` gets one sum per workgroup (the return value = how many; <= GOM_LPIPS_HEAD_BLOCKS per image,
// image b's at block_sums + b * stride); gom_lpips_fold_values turns the five taps' sums into the caller's value rows.  C = 64, 128, 256 or 512.
int gom_lpips_layer_backward_value_planes(int B, int C, int HW, const void *f0, const void *f1, const float *w, const float *grad_out, void *d_f0, float *block_sums,
                                          int *n_blocks, size_t f_lo, size_t d_lo, const void *pool_dy, size_t pool_dy_lo, int W, void *stream) {
    if (C != 64 && C != 128 && C != 256 && C != 512) { gom_set_error("LPIPS head (backward + value): C must be 64, 128, 256 or 512"); return -1; }
    if (B <= 0 || HW <= 0 || !block_sums || !n_blocks) { gom_set_error("LPIPS head (backward + value): bad arguments"); return -1; }
    // a workgroup per 16 pixels -- or, without a pool to route (the last tap: 32 x 32 pixels x 512 channels at 512^2), one PIXEL per group of C / 8 lanes:
    // 64 workgroups walking four pixels each one behind the other were a 12 us latency chain on a quarter of the CUs
    const size_t groups = pool_dy ? ((size_t)HW + 15) / 16 : ((size_t)HW * (size_t)(C / 8) + 255) / 256;
    const dim3 gridb((unsigned)(groups < GOM_LPIPS_HEAD_BLOCKS ? groups : GOM_LPIPS_HEAD_BLOCKS), B);
    *n_blocks = (int)gridb.x;
    if (pool_dy && (W <= 0 || (W & 1) || HW % W || ((HW / W) & 1))) { gom_set_error("LPIPS head (backward + value + pool): even image sides"); return -1; }
#define GOM_HEADV(LPP_, POOL_) hipLaunchKernelGGL((k_lpips_head_nhwc_1p<true, LPP_, true, POOL_>), gridb, dim3(256), 0, (hipStream_t)stream, (size_t)HW, (const bf16_t *)f0, \
                                                  (const bf16_t *)f1, w, grad_out, block_sums, (bf16_t *)d_f0, f_lo, d_lo, (const bf16_t *)pool_dy, pool_dy_lo, W)
    if (pool_dy) { if (C == 64) GOM_HEADV(8, true); else if (C == 128) GOM_HEADV(16, true); else if (C == 256) GOM_HEADV(32, true); else GOM_HEADV(64, true); }
    else { if (C == 64) GOM_HEADV(8, false); else if (C == 128) GOM_HEADV(16, false); else if (C == 256) GOM_HEADV(32, false); else GOM_HEADV(64, false); }
#undef GOM_HEADV
    GOM_LAUNCH_CHECK();
    return 0;
}
int gom_lpips_fold_values(int B, const float *block_sums, const int *n_blocks5, float *partials, void *stream) {
    hipLaunchKernelGGL(k_lpips_fold_values, dim3(5, B), dim3(256), 0, (hipStream_t)stream, B, block_sums, GOM_LPIPS_HEAD_BLOCKS, n_blocks5[0], n_blocks5[1], n_blocks5[2], n_blocks5[3],
                       n_blocks5[4], partials);
    GOM_LAUNCH_CHECK();
    return 0;
}
int gom_lpips_layer_backward_planes(int B, int C, int HW, const void *f0, const void *f1, const float *w, const float *grad_out, void *d_f0, size_t f_lo, size_t d_lo,
                                    void *stream) {
    if (f_lo && C != 64 && C != 128 && C != 256 && C != 512) { gom_set_error("bf16x3 LPIPS head: C must be 64, 128, 256 or 512"); return -1; }
    if (B <= 0 || C <= 0 || (C % 128 && C != 64) || HW <= 0) { gom_set_error("gom_lpips_layer_backward_nhwc_bf16: C must be 64 or a multiple of 128"); return -1; }
    const size_t groups = ((size_t)HW + 15) / 16;
    const dim3 gridb((unsigned)(groups < 4096 ? groups : 4096), B);
    if (C == 64) GOM_HEAD1P(true, 8, gridb, 256, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, grad_out, nullptr, (bf16_t *)d_f0, f_lo, d_lo);
    else if (C == 128) GOM_HEAD1P(true, 16, gridb, 256, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, grad_out, nullptr, (bf16_t *)d_f0, f_lo, d_lo);
    else if (C == 256) GOM_HEAD1P(true, 32, gridb, 256, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, grad_out, nullptr, (bf16_t *)d_f0, f_lo, d_lo);
    else if (C == 512) GOM_HEAD1P(true, 64, gridb, 256, (size_t)HW, (const bf16_t *)f0, (const bf16_t *)f1, w, grad_out, nullptr, (bf16_t *)d_f0, f_lo, d_lo);
    else
    hipLaunchKernelGGL(k_lpips_head_nhwc<true>, dim3((unsigned)(groups < 4096 ? groups : 4096), B), dim3(256), 0, (hipStream_t)stream, C, (size_t)HW, (const bf16_t *)f0,
                       (const bf16_t *)f1, w, grad_out, nullptr, (bf16_t *)d_f0);
    GOM_LAUNCH_CHECK();
    return 0;
}
