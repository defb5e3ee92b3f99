// ds_read_b64_tr_b16 on a row-major LDS image that also serves 16-byte row reads: pins the lane map of the transposed
// read and the chunk swizzle used by the weight-gradient products of mgn_train.hip.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/tr_read scripts/micro/tr_read.hip && scripts/micro/tr_read
// One wave computes D = G^T A for a 32 x 128 tile of small integers in bf16 (exact) two ways -- operands fetched with
// transposed reads from the row-major image -- and the host compares with the plain triple loop.  It also reads every
// row back with 16-byte reads.  Prints "ok" or the first mismatches.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int H = 128, TM = 32;

__device__ __forceinline__ int swz_off(int row, int ch) { return row * H + 8 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ __forceinline__ u32x2 lds_tr4(const unsigned short* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

// rows 16s + 8h + 0..7 of column 32*blk + (lane & 31): the 32x32x16 operand of lane (j = lane & 31, h = lane >> 5)
__device__ __forceinline__ u32x4 tr_operand(const unsigned short* img, int lane, int s, int blk) {
    const int h = lane >> 5, gg = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int ch = 4 * blk + 2 * gg + (p >> 1);
    const u32x2 a = lds_tr4(img + swz_off(16 * s + 8 * h + q, ch) + 4 * (p & 1));
    const u32x2 b = lds_tr4(img + swz_off(16 * s + 8 * h + 4 + q, ch) + 4 * (p & 1));
    return u32x4{a[0], a[1], b[0], b[1]};
}

__global__ __launch_bounds__(64) void k(const unsigned short* G, const unsigned short* A, float* D, unsigned short* rows_out) {
    __shared__ __attribute__((aligned(16))) unsigned short Gs[TM * H], As[TM * H];
    const int lane = threadIdx.x;
    for (int i = lane; i < TM * H / 4; i += 64) {           // 4 halves per step, as the kernels store them
        const int row = i / (H / 4), k = 4 * (i % (H / 4));
        *(u32x2*)(Gs + swz_off(row, k >> 3) + (k & 4)) = *(const u32x2*)(G + row * H + k);
        *(u32x2*)(As + swz_off(row, k >> 3) + (k & 4)) = *(const u32x2*)(A + row * H + k);
    }
    __syncthreads();
    const int j = lane & 31, h = lane >> 5;
    for (int w = 0; w < 4; ++w)
        for (int kb = 0; kb < 4; ++kb) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < 2; ++s) {
                const u32x4 g = tr_operand(Gs, lane, s, w), a = tr_operand(As, lane, s, kb);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
            }
            for (int reg = 0; reg < 16; ++reg) {
                const int f = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                D[f * H + 32 * kb + j] = acc[reg];
            }
        }
    for (int c = 0; c < 8; ++c)                               // row reads: row j, halves 16c + 8h .. + 7
        *(u32x4*)(rows_out + j * H + 16 * c + 8 * h) = *(const u32x4*)(Gs + swz_off(j, 2 * c + h));
}

static unsigned short bf16_of(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }

int main() {
    std::vector<unsigned short> G(TM * H), A(TM * H);
    std::vector<float> Gf(TM * H), Af(TM * H);
    unsigned seed = 12345u;
    auto next = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((int)((seed >> 20) & 15) - 8); };
    for (int i = 0; i < TM * H; ++i) { Gf[i] = next(); Af[i] = next(); G[i] = bf16_of(Gf[i]); A[i] = bf16_of(Af[i]); }
    unsigned short *dG, *dA, *dR;
    float* dD;
    if (hipMalloc(&dG, TM * H * 2) != hipSuccess || hipMalloc(&dA, TM * H * 2) != hipSuccess ||
        hipMalloc(&dR, TM * H * 2) != hipSuccess || hipMalloc(&dD, H * H * 4) != hipSuccess) { printf("alloc failed\n"); return 2; }
    (void)hipMemcpy(dG, G.data(), TM * H * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dA, A.data(), TM * H * 2, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dG, dA, dD, dR);
    std::vector<float> D(H * H);
    std::vector<unsigned short> R(TM * H);
    if (hipMemcpy(D.data(), dD, H * H * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
    (void)hipMemcpy(R.data(), dR, TM * H * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int f = 0; f < H; ++f)
        for (int c = 0; c < H; ++c) {
            float ref = 0.f;
            for (int e = 0; e < TM; ++e) ref += Gf[e * H + f] * Af[e * H + c];
            if (ref != D[f * H + c] && bad++ < 8) printf("D[%d][%d] = %g, expected %g\n", f, c, D[f * H + c], ref);
        }
    int badr = 0;
    for (int i = 0; i < TM * H; ++i)
        if (R[i] != G[i] && badr++ < 8) printf("row read [%d][%d] = %04x, expected %04x\n", i / H, i % H, R[i], G[i]);
    printf("transposed-read product: %s (%d mismatches); row reads: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad,
           badr ? "WRONG" : "ok", badr);
    return bad || badr ? 1 : 0;
}
