// valu_microbench.hip -- what one instruction of each class costs a gfx950 SIMD in steady state, at 1, 2,
// 4, 5 and 8 resident waves per SIMD.  Answers the question the round-2 review left open:
// is a wave64 VALU instruction 4 cycles of its SIMD (what bench.py's valu_issue assumed) or 2, does a packed
// v_pk_fma_f32 cost one plain instruction or two, and do scalar instructions share the vector ones' issue
// slots?  The render kernels' issue ceiling (bench.py) is priced with these numbers.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o build/valu_microbench
//   build/valu_microbench [out.csv] [class-name substring]
//
// Method: every wave runs a stream of 32 independent instructions of one class (inline asm on distinct
// registers: no dependency chain, nothing for the compiler to fold) over and over for a FIXED WINDOW of
// shader-clock ticks (s_memtime) and counts how many it got through.  A block is four waves -- one per
// SIMD of its CU -- and grids of (CUs x 1..8) blocks are launched; the dispatcher does not spread blocks evenly
// over the CUs, so every wave records the SIMD it ran on (HW_ID, XCC_ID) and the SIMDs are grouped by how many
// waves they actually held for the whole window.  SIMD cycles per instruction = window / (instructions the
// SIMD's waves executed).  The real shader clock in the window comes from the constant 100 MHz counter read
// beside s_memtime.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr unsigned long long kWindow = 3000000ull;   // ticks (~1.4 ms)
constexpr int kInner = 8;                            // bodies between two looks at the clock

typedef float f2 __attribute__((ext_vector_type(2)));

enum Kind {
    kFma, kAdd, kMul, kPkFma, kPkMul, kPkAdd, kMax, kMin, kMax3, kCndmask, kCmp, kCmpVcc, kSqrt, kRcp, kMulLo, kMulHi,
    kAddU32, kAnd, kLshl, kCvt, kReadlane, kMov, kFmaF64, kMaxI32, kMax3I32, kMed3, kCmpU32, kSubRev, kOr3, kBfe, kMulU24, kFmac, kPkAddU,
    kSalu, kSmem, kLdsRead, kLdsWrite,
    kFmaSalu11, kFmaSalu21, kPkFmaSalu11, kFmaSmem41, kFmaPk11, kFmaMax11, kFmaCnd11, kFmaPk31, kFmaPkRuns, kFmaAddU11, kPkMax11,
    kIeeeDiv, kIeeeSqrt, kNKinds
};
struct KindInfo { const char *name; int per_body; };   // instructions of the stream per 32-slot body
static const KindInfo kInfo[kNKinds] = {
    {"v_fma_f32", 32}, {"v_add_f32", 32}, {"v_mul_f32", 32}, {"v_pk_fma_f32", 32}, {"v_pk_mul_f32", 32}, {"v_pk_add_f32", 32},
    {"v_max_f32", 32}, {"v_min_f32", 32}, {"v_max3_f32", 32}, {"v_cndmask_b32 (sgpr mask)", 32}, {"v_cmp_gt_f32 -> sgpr pair", 32},
    {"v_cmp_gt_f32 -> vcc", 32}, {"v_sqrt_f32", 32}, {"v_rcp_f32", 32}, {"v_mul_lo_u32", 32}, {"v_mul_hi_u32", 32},
    {"v_add_u32", 32}, {"v_and_b32", 32}, {"v_lshrrev_b32", 32}, {"v_cvt_f32_u32", 32}, {"v_readlane_b32", 32}, {"v_mov_b32", 32},
    {"v_fma_f64", 32}, {"v_max_i32", 32}, {"v_max3_i32", 32}, {"v_med3_f32", 32}, {"v_cmp_lt_u32 -> vcc", 32}, {"v_subrev_f32", 32},
    {"v_or3_b32", 32}, {"v_bfe_u32", 32}, {"v_mul_u32_u24", 32}, {"v_fmac_f32", 32}, {"v_pk_add_u16", 32},
    {"s_add_u32", 32}, {"s_load_dwordx4 (L1-hot, 8 in flight)", 32}, {"ds_read_b32", 32}, {"ds_write_b32", 32},
    {"mix 1 v_fma_f32 : 1 s_add_u32 (per instruction)", 32}, {"mix 2 v_fma_f32 : 1 s_add_u32 (per instruction)", 30},
    {"mix 1 v_pk_fma_f32 : 1 s_add_u32 (per instruction)", 32}, {"mix 4 v_fma_f32 : 1 s_load_dwordx4 (per instruction)", 30},
    {"mix 1 v_fma_f32 : 1 v_pk_fma_f32 (per instruction)", 32}, {"mix 1 v_fma_f32 : 1 v_max_f32 (per instruction)", 32},
    {"mix 1 v_fma_f32 : 1 v_cndmask_b32 (per instruction)", 32}, {"mix 3 v_fma_f32 : 1 v_pk_fma_f32 (per instruction)", 32},
    {"mix runs of 16 v_fma_f32, 16 v_pk_fma_f32 (per instruction)", 32}, {"mix 1 v_fma_f32 : 1 v_add_u32 (per instruction)", 32},
    {"mix 1 v_pk_fma_f32 : 1 v_max_f32 (per instruction)", 32},
    {"IEEE a / b (per result)", 32}, {"IEEE sqrtf (per result)", 32}};

// One asm statement per body: 32 instruction lines over 16 registers (each register twice, 16 lines apart), so
// the compiler's hazard recogniser has nothing to pad between statements.
#define LN(pre, r, post) pre " %" #r post "\n\t"
#define L16(pre, post) LN(pre, 0, post) LN(pre, 1, post) LN(pre, 2, post) LN(pre, 3, post) LN(pre, 4, post) LN(pre, 5, post) \
                       LN(pre, 6, post) LN(pre, 7, post) LN(pre, 8, post) LN(pre, 9, post) LN(pre, 10, post) LN(pre, 11, post) \
                       LN(pre, 12, post) LN(pre, 13, post) LN(pre, 14, post) LN(pre, 15, post)
#define L32(pre, post) L16(pre, post) L16(pre, post)
// rr: "op %r, %r" + post
#define RR(op, r, post) op " %" #r ", %" #r post "\n\t"
#define R16(op, post) RR(op, 0, post) RR(op, 1, post) RR(op, 2, post) RR(op, 3, post) RR(op, 4, post) RR(op, 5, post) RR(op, 6, post) \
                      RR(op, 7, post) RR(op, 8, post) RR(op, 9, post) RR(op, 10, post) RR(op, 11, post) RR(op, 12, post) RR(op, 13, post) \
                      RR(op, 14, post) RR(op, 15, post)
#define R32(op, post) R16(op, post) R16(op, post)
// mixes: a vector line on register r followed by a scalar line
#define MX(op, r, post, sline) op " %" #r ", %" #r post "\n\t" sline "\n\t"
#define MX16(op, post, s_a, s_b) MX(op, 0, post, s_a) MX(op, 1, post, s_b) MX(op, 2, post, s_a) MX(op, 3, post, s_b) MX(op, 4, post, s_a) \
                                 MX(op, 5, post, s_b) MX(op, 6, post, s_a) MX(op, 7, post, s_b) MX(op, 8, post, s_a) MX(op, 9, post, s_b) \
                                 MX(op, 10, post, s_a) MX(op, 11, post, s_b) MX(op, 12, post, s_a) MX(op, 13, post, s_b) MX(op, 14, post, s_a) \
                                 MX(op, 15, post, s_b)
#define A16(v) "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), \
               "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(float *out, unsigned long long *count, unsigned long long *wall, const float *tab, float seed)
{
    constexpr bool kUsesP = KIND == kFmaPk31 || KIND == kFmaPkRuns || KIND == kPkMax11 || KIND == kPkFma || KIND == kPkMul || KIND == kPkAdd || KIND == kPkFmaSalu11 || KIND == kFmaPk11;
    constexpr bool kUsesU = KIND == kMaxI32 || KIND == kMax3I32 || KIND == kCmpU32 || KIND == kOr3 || KIND == kBfe || KIND == kMulU24 || KIND == kPkAddU || KIND == kFmaAddU11 || KIND == kMulLo || KIND == kMulHi || KIND == kAddU32 || KIND == kReadlane || KIND == kAnd || KIND == kLshl;
    constexpr bool kUsesD = KIND == kFmaF64;
    constexpr bool kUsesA = !kUsesP && !kUsesU && !kUsesD || KIND == kFmaPk11 || KIND == kFmaPk31 || KIND == kFmaPkRuns || KIND == kPkMax11 || KIND == kFmaAddU11;
    __shared__ float lds[256 * 4];
    float a[16], a2[2] = {seed, seed + 1.0f};
    f2 p[16];
    double d[16];
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
        u[i] = (unsigned)(threadIdx.x * 2654435761u + i);
        p[i] = f2{a[i], a[i] + 1.0f};
        d[i] = (double)a[i];
    }
    lds[threadIdx.x] = seed;
    const float x = seed * 0.5f + 1.0f, y = seed * 0.25f + 1e-6f;
    const f2 px = f2{x, x}, py = f2{y, y};
    const double xd = (double)x, yd = (double)y;
    unsigned long long mask = __ballot(a[0] > 1.0005f);
    unsigned s0 = 1, s1 = 2;
    const unsigned long long tabaddr = (unsigned long long)tab;
    const unsigned lp = (unsigned)(unsigned long long)&lds[threadIdx.x];
    unsigned long long n = 0;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (;;) {
        for (int it = 0; it < kInner; it++) {
            if (KIND == kFma)          asm volatile(R32("v_fma_f32", ", %16, %17") : A16(a) : "v"(x), "v"(y));
            else if (KIND == kAdd)     asm volatile(R32("v_add_f32", ", %16") : A16(a) : "v"(y));
            else if (KIND == kMul)     asm volatile(R32("v_mul_f32", ", %16") : A16(a) : "v"(x));
            else if (KIND == kPkFma)   asm volatile(R32("v_pk_fma_f32", ", %16, %17") : A16(p) : "v"(px), "v"(py));
            else if (KIND == kPkMul)   asm volatile(R32("v_pk_mul_f32", ", %16") : A16(p) : "v"(px));
            else if (KIND == kPkAdd)   asm volatile(R32("v_pk_add_f32", ", %16") : A16(p) : "v"(py));
            else if (KIND == kMax)     asm volatile(R32("v_max_f32", ", %16") : A16(a) : "v"(y));
            else if (KIND == kMin)     asm volatile(R32("v_min_f32", ", %16") : A16(a) : "v"(x));
            else if (KIND == kMax3)    asm volatile(R32("v_max3_f32", ", %16, %17") : A16(a) : "v"(x), "v"(y));
            else if (KIND == kCndmask) asm volatile(R32("v_cndmask_b32", ", %16, %17") : A16(a) : "v"(y), "s"(mask));
            else if (KIND == kCmp)     asm volatile(L32("v_cmp_gt_f32 s[20:21],", ", %16") : A16(a) : "v"(y) : "s20", "s21");
            else if (KIND == kCmpVcc)  asm volatile(L32("v_cmp_gt_f32 vcc,", ", %16") : A16(a) : "v"(y) : "vcc");
            else if (KIND == kSqrt)    asm volatile(R32("v_sqrt_f32", "") : A16(a));
            else if (KIND == kRcp)     asm volatile(R32("v_rcp_f32", "") : A16(a));
            else if (KIND == kMulLo)   asm volatile(R32("v_mul_lo_u32", ", %16") : A16(u) : "v"(0xD2511F53u));
            else if (KIND == kMulHi)   asm volatile(R32("v_mul_hi_u32", ", %16") : A16(u) : "v"(0xCD9E8D57u));
            else if (KIND == kAddU32)  asm volatile(R32("v_add_u32", ", %16") : A16(u) : "v"(0x9E3779B9u));
            else if (KIND == kAnd)     asm volatile(R32("v_and_b32", ", %16") : A16(u) : "v"(0xfffffff3u));
            else if (KIND == kLshl)    asm volatile(R32("v_lshrrev_b32", ", %16") : A16(u) : "v"(0x7fu));   // %r = %16 >> %r
            else if (KIND == kCvt)     asm volatile(R32("v_cvt_f32_u32", "") : A16(a));
            else if (KIND == kReadlane) asm volatile(L32("v_readlane_b32 s20,", ", 5") : A16(u) : : "s20");
            else if (KIND == kMov)     asm volatile(L32("v_mov_b32", ", %16") : A16(a) : "v"(x));
            else if (KIND == kFmaF64)  asm volatile(R32("v_fma_f64", ", %16, %17") : A16(d) : "v"(xd), "v"(yd));
            else if (KIND == kMaxI32)  asm volatile(R32("v_max_i32", ", %16") : A16(u) : "v"(0x3f800000u));
            else if (KIND == kMax3I32) asm volatile(R32("v_max3_i32", ", %16, %17") : A16(u) : "v"(0x3f800000u), "v"(0x40000000u));
            else if (KIND == kMed3)    asm volatile(R32("v_med3_f32", ", %16, %17") : A16(a) : "v"(x), "v"(y));
            else if (KIND == kCmpU32)  asm volatile(L32("v_cmp_lt_u32 vcc,", ", %16") : A16(u) : "v"(0x3f800000u) : "vcc");
            else if (KIND == kSubRev)  asm volatile(R32("v_subrev_f32", ", %16") : A16(a) : "v"(y));
            else if (KIND == kOr3)     asm volatile(R32("v_or3_b32", ", %16, %17") : A16(u) : "v"(0x10u), "v"(0x3u));
            else if (KIND == kBfe)     asm volatile(R32("v_bfe_u32", ", 2, 20") : A16(u));
            else if (KIND == kMulU24)  asm volatile(R32("v_mul_u32_u24", ", %16") : A16(u) : "v"(0x51F53u));
            else if (KIND == kFmac)    asm volatile(L32("v_fmac_f32", ", %16, %17") : A16(a) : "v"(x), "v"(y));
            else if (KIND == kPkAddU)  asm volatile(R32("v_pk_add_u16", ", %16") : A16(u) : "v"(0x00010001u));
            else if (KIND == kFmaMax11) { asm volatile(MX16("v_fma_f32", ", %18, %19", "v_max_f32 %16, %16, %19", "v_max_f32 %17, %17, %19") : A16(a), "+v"(a2[0]), "+v"(a2[1]) : "v"(x), "v"(y)); }
            else if (KIND == kFmaCnd11) { asm volatile(MX16("v_fma_f32", ", %18, %19", "v_cndmask_b32 %16, %16, %19, %20", "v_cndmask_b32 %17, %17, %19, %20") : A16(a), "+v"(a2[0]), "+v"(a2[1]) : "v"(x), "v"(y), "s"(mask)); }
            else if (KIND == kFmaPk31) {
#define Q4(r0, r1, r2, pr) RR("v_fma_f32", r0, ", %20, %21") RR("v_fma_f32", r1, ", %20, %21") RR("v_fma_f32", r2, ", %20, %21") "v_pk_fma_f32 %" #pr ", %" #pr ", %22, %23\n\t"
                asm volatile(Q4(0, 1, 2, 16) Q4(3, 4, 5, 17) Q4(6, 7, 8, 18) Q4(9, 10, 11, 19) Q4(12, 13, 14, 16) Q4(15, 0, 1, 17) Q4(2, 3, 4, 18) Q4(5, 6, 7, 19)
                             : A16(a), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(x), "v"(y), "v"(px), "v"(py));
#undef Q4
            } else if (KIND == kFmaPkRuns) {
                asm volatile(R16("v_fma_f32", ", %16, %17") : A16(a) : "v"(x), "v"(y));
                asm volatile(R16("v_pk_fma_f32", ", %16, %17") : A16(p) : "v"(px), "v"(py));
            } else if (KIND == kFmaAddU11) { asm volatile(MX16("v_fma_f32", ", %18, %19", "v_add_u32 %16, %16, %20", "v_add_u32 %17, %17, %20") : A16(a), "+v"(u[0]), "+v"(u[1]) : "v"(x), "v"(y), "v"(0x9E3779B9u)); }
            else if (KIND == kPkMax11) { asm volatile(MX16("v_pk_fma_f32", ", %18, %19", "v_max_f32 %16, %16, %20", "v_max_f32 %17, %17, %20") : A16(p), "+v"(a[0]), "+v"(a[1]) : "v"(px), "v"(py), "v"(y)); }
            else if (KIND == kSalu) {
#define S4 "s_add_u32 %0, %0, 7\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 s20, s20, 3\n\ts_add_u32 s21, s21, 1\n\t"
                asm volatile(S4 S4 S4 S4 S4 S4 S4 S4 : "+s"(s0), "+s"(s1) : : "s20", "s21", "scc");
#undef S4
            } else if (KIND == kSmem) {
                // eight loads in flight, then one wait: four rounds per body
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 r0, r1, r2, r3, r4, r5, r6, r7;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    asm volatile("s_load_dwordx4 %0, %8, 0x0\n\ts_load_dwordx4 %1, %8, 0x10\n\ts_load_dwordx4 %2, %8, 0x20\n\ts_load_dwordx4 %3, %8, 0x30\n\t"
                                 "s_load_dwordx4 %4, %8, 0x40\n\ts_load_dwordx4 %5, %8, 0x50\n\ts_load_dwordx4 %6, %8, 0x60\n\ts_load_dwordx4 %7, %8, 0x70\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4), "=&s"(r5), "=&s"(r6), "=&s"(r7) : "s"(tabaddr) : "memory");
                    s0 ^= r0.x ^ r7.w;
                }
            } else if (KIND == kLdsRead)  asm volatile(L32("ds_read_b32", ", %16") "s_waitcnt lgkmcnt(0)" : A16(a) : "v"(lp) : "memory");
            else if (KIND == kLdsWrite)   asm volatile(L32("ds_write_b32 %16,", "") "s_waitcnt lgkmcnt(0)" : A16(a) : "v"(lp) : "memory");
            else if (KIND == kFmaSalu11)  asm volatile(MX16("v_fma_f32", ", %16, %17", "s_add_u32 %18, %18, 7", "s_add_u32 %19, %19, 5")
                                                       : A16(a) : "v"(x), "v"(y), "s"(s0), "s"(s1) : "scc");
            else if (KIND == kFmaSalu21) {
                // 30 instructions: v v s, ten times
#define T3(r0, r1, sl) RR("v_fma_f32", r0, ", %16, %17") RR("v_fma_f32", r1, ", %16, %17") sl "\n\t"
                asm volatile(T3(0, 1, "s_add_u32 %18, %18, 7") T3(2, 3, "s_add_u32 %19, %19, 5") T3(4, 5, "s_add_u32 %18, %18, 7") T3(6, 7, "s_add_u32 %19, %19, 5")
                             T3(8, 9, "s_add_u32 %18, %18, 7") T3(10, 11, "s_add_u32 %19, %19, 5") T3(12, 13, "s_add_u32 %18, %18, 7") T3(14, 15, "s_add_u32 %19, %19, 5")
                             T3(0, 1, "s_add_u32 %18, %18, 7") T3(2, 3, "s_add_u32 %19, %19, 5")
                             : A16(a) : "v"(x), "v"(y), "s"(s0), "s"(s1) : "scc");
#undef T3
            } else if (KIND == kPkFmaSalu11) asm volatile(MX16("v_pk_fma_f32", ", %16, %17", "s_add_u32 %18, %18, 7", "s_add_u32 %19, %19, 5")
                                                          : A16(p) : "v"(px), "v"(py), "s"(s0), "s"(s1) : "scc");
            else if (KIND == kFmaSmem41) {
                // 30 instructions: six groups of four v_fma_f32 and one s_load_dwordx4, one wait at the end
#define V4(r0, r1, r2, r3) RR("v_fma_f32", r0, ", %16, %17") RR("v_fma_f32", r1, ", %16, %17") RR("v_fma_f32", r2, ", %16, %17") RR("v_fma_f32", r3, ", %16, %17")
                asm volatile("s_load_dwordx4 s[20:23], %18, 0x0\n\t" V4(0, 1, 2, 3) "s_load_dwordx4 s[24:27], %18, 0x10\n\t" V4(4, 5, 6, 7)
                             "s_load_dwordx4 s[28:31], %18, 0x20\n\t" V4(8, 9, 10, 11) "s_load_dwordx4 s[44:47], %18, 0x30\n\t" V4(12, 13, 14, 15)
                             "s_load_dwordx4 s[36:39], %18, 0x40\n\t" V4(0, 1, 2, 3) "s_load_dwordx4 s[40:43], %18, 0x50\n\t" V4(4, 5, 6, 7)
                             "s_waitcnt lgkmcnt(0)"
                             : A16(a) : "v"(x), "v"(y), "s"(tabaddr)
                             : "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s44", "s45", "s46", "s47",
                               "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43");
#undef V4
            } else if (KIND == kFmaPk11) {
#define FP(r, pr) RR("v_fma_f32", r, ", %20, %21") "v_pk_fma_f32 %" #pr ", %" #pr ", %22, %23\n\t"
                asm volatile(FP(0, 16) FP(1, 17) FP(2, 18) FP(3, 19) FP(4, 16) FP(5, 17) FP(6, 18) FP(7, 19) FP(8, 16) FP(9, 17) FP(10, 18) FP(11, 19)
                             FP(12, 16) FP(13, 17) FP(14, 18) FP(15, 19)
                             : A16(a), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(x), "v"(y), "v"(px), "v"(py));
#undef FP
            } else if (KIND == kIeeeDiv) {
#pragma unroll
                for (int i = 0; i < 16; i++) { a[i] = x / a[i]; asm volatile("" : "+v"(a[i])); }
#pragma unroll
                for (int i = 0; i < 16; i++) { a[i] = x / a[i]; asm volatile("" : "+v"(a[i])); }
            } else if (KIND == kIeeeSqrt) {
#pragma unroll
                for (int i = 0; i < 16; i++) { a[i] = sqrtf(a[i]); asm volatile("" : "+v"(a[i])); }
#pragma unroll
                for (int i = 0; i < 16; i++) { a[i] = sqrtf(a[i]); asm volatile("" : "+v"(a[i])); }
            }
        }
        n += kInner;
        if (__builtin_amdgcn_s_memtime() - t0 >= kWindow)
            break;
    }
    const unsigned long long w1 = wall_clock64();
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (kUsesA) acc += a[i];
        if (kUsesU) acc += (float)u[i];
        if (kUsesP) acc += p[i].x + p[i].y;
        if (kUsesD) acc += (float)d[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + a2[0] + a2[1] + (float)(s0 + s1) + (float)mask + lds[(threadIdx.x + 1) & 255];
    if ((threadIdx.x & 63) == 0) {
        // where the wave ran: HW_ID (wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) and XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        count[w] = n;
        wall[3 * w] = w0;
        wall[3 * w + 1] = w1;
        wall[3 * w + 2] = ((unsigned long long)(xcc & 15u) << 16) | (hw & 0xff30u);
    }
}

// Waves are grouped by the SIMD they actually ran on (the dispatcher does not spread blocks evenly), and SIMDs by how
// many waves they held: cycles per instruction of a SIMD = window / instructions its waves executed.
template <int KIND>
static void run(int cus, float *out, unsigned long long *count, unsigned long long *wall, const float *tab,
                std::vector<unsigned long long> &h, std::vector<unsigned long long> &hw, FILE *csv)
{
    int resident = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_bench<KIND>, 256, 0));
    std::map<int, std::pair<double, double>> by_w;      // waves on the SIMD -> (SIMDs, summed cycles per instruction)
    double mhz_sum = 0;
    int mhz_n = 0;
    for (int W : {1, 2, 3, 4, 5, 6, 8}) {
        if (W > resident) continue;
        const int blocks = cus * W, waves = blocks * 4;
        k_bench<KIND><<<blocks, 256>>>(out, count, wall, tab, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), count, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hw.data(), wall, sizeof(unsigned long long) * waves * 3, hipMemcpyDeviceToHost));
        // only waves whose window overlaps every other wave's on the same SIMD count: take the SIMDs whose waves all
        // started within 2 % of the window of each other
        std::map<unsigned long long, std::vector<int>> simd;
        for (int i = 0; i < waves; i++) simd[hw[3 * i + 2]].push_back(i);
        for (auto &kv : simd) {
            unsigned long long lo = ~0ull, hi = 0, dur = 0;
            double bodies = 0;
            for (int i : kv.second) {
                lo = std::min(lo, hw[3 * i]); hi = std::max(hi, hw[3 * i]);
                dur += hw[3 * i + 1] - hw[3 * i];
                bodies += (double)h[i];
            }
            const double mean_dur = (double)dur / kv.second.size();
            if ((double)(hi - lo) > 0.02 * mean_dur) continue;
            auto &acc = by_w[(int)kv.second.size()];
            acc.first += 1.0;
            acc.second += (double)kWindow / (bodies * kInfo[KIND].per_body);
            mhz_sum += (double)kWindow / (mean_dur / 100.0);
            mhz_n++;
        }
    }
    const double mhz = mhz_n ? mhz_sum / mhz_n : 0.0;
    std::printf("%-52s", kInfo[KIND].name);
    for (auto &kv : by_w) {
        if (kv.second.first < 16) continue;              // too few SIMDs with that many waves to say
        const double cyc = kv.second.second / kv.second.first;
        std::printf("  W=%d %6.3f", kv.first, cyc);
        if (csv) std::fprintf(csv, "\"%s\",%d,%.4f,%.0f,%.0f\n", kInfo[KIND].name, kv.first, cyc, kv.second.first, mhz);
    }
    std::printf("   cycles/inst/SIMD (clock %4.0f MHz)\n", mhz);
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs, nominal clock %d kHz, window %llu ticks\n", prop.gcnArchName, cus, prop.clockRate, kWindow);
    float *out, *tab;
    unsigned long long *count, *wall;
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&out, sizeof(float) * max_blocks * 256));
    CHECK(hipMalloc(&tab, 4096));
    CHECK(hipMemset(tab, 0, 4096));
    CHECK(hipMalloc(&count, sizeof(unsigned long long) * max_blocks * 4));
    CHECK(hipMalloc(&wall, sizeof(unsigned long long) * max_blocks * 4 * 3));
    std::vector<unsigned long long> h((size_t)max_blocks * 4), hw((size_t)max_blocks * 4 * 3);
    FILE *csv = argc > 1 ? std::fopen(argv[1], "w") : nullptr;
    if (csv) std::fprintf(csv, "class,waves_on_simd,simd_cycles_per_inst,simds_sampled,shader_mhz\n");
    const char *only = argc > 2 ? argv[2] : nullptr;
#define RUN(K) if (!only || std::strstr(kInfo[K].name, only)) run<K>(cus, out, count, wall, tab, h, hw, csv);
    RUN(kFma) RUN(kAdd) RUN(kMul) RUN(kPkFma) RUN(kPkMul) RUN(kPkAdd) RUN(kMax) RUN(kMin) RUN(kMax3) RUN(kCndmask) RUN(kCmp) RUN(kCmpVcc)
    RUN(kSqrt) RUN(kRcp) RUN(kMulLo) RUN(kMulHi) RUN(kAddU32) RUN(kAnd) RUN(kLshl) RUN(kCvt) RUN(kReadlane) RUN(kMov) RUN(kFmaF64) RUN(kMaxI32) RUN(kMax3I32) RUN(kMed3) RUN(kCmpU32) RUN(kSubRev) RUN(kOr3) RUN(kBfe) RUN(kMulU24) RUN(kFmac) RUN(kPkAddU)
    RUN(kSalu) RUN(kSmem) RUN(kLdsRead) RUN(kLdsWrite)
    RUN(kFmaSalu11) RUN(kFmaSalu21) RUN(kPkFmaSalu11) RUN(kFmaSmem41) RUN(kFmaPk11) RUN(kFmaMax11) RUN(kFmaCnd11) RUN(kFmaPk31) RUN(kFmaPkRuns) RUN(kFmaAddU11) RUN(kPkMax11)
    RUN(kIeeeDiv) RUN(kIeeeSqrt)
    if (csv) std::fclose(csv);
    return 0;
}
