// The 256x256x64 bf16 GEMM as FOUR waves of 128 x 128 per workgroup - one wave per SIMD with the whole 512-register file - beside the
// eight waves of 128 x 64 of gemm256.hip.  gemm_wide256w4_kernel<EPI> is the product kernel (persistent tiles, the epilogues of
// gemm256.hip; hwocr_gemm_wide256_w4 says for which shapes it is taken); the kernels under HWOCR_DIAG at the end are the experiment it
// grew out of (tools/bench_gemm_w4.py: timing ablations, the 32x32x16 form, in-kernel cycle stamps).
//
// Why (a first four-wave form lost to the eight-wave kernel in round 1): the eight-wave main loop is 2566 shader
// cycles per K tile against 2048 of matrix-pipe time at every CU count, under a clock the power budget sets
// (profiles/r03z_gemm_cus.txt), and both the cycles and the power point at the LDS: a wave of 128 x 64 reads (128 + 64) fragment rows
// per 8192 outputs, a wave of 128 x 128 reads (128 + 128) per 16384 - a third fewer LDS bytes per FLOP.  The vendor library's kernel
// for these shapes is exactly that shape (dispatch records: 256 threads, 130 KiB LDS), and its loop is one MFMA per slot with at
// most one LDS read or one LDS-DMA behind it - which is how this loop is written: program order IS issue order
// (__builtin_amdgcn_sched_barrier(0) after every slot).
//
// Per K tile t (64 wide; stage t & 1 of two 64-KiB stages; fragments of its two 32-wide k-steps in two register sets):
//   first half : 64 MFMAs on the k-step-0 fragments; behind the first 16 the reads of the k-step-1 fragments, then a wait + barrier
//                (every wave has everything of tile t in registers: stage t & 1 is free), then DMAs of tile t + 2 into it;
//   second half: 64 MFMAs on the k-step-1 fragments; behind them the rest of tile t + 2's 16 DMAs (one per six slots), a counted wait
//                + barrier (tile t + 1 has landed), then the 16 reads of tile t + 1's k-step-0 fragments.
// LDS image, swizzle and DMA lane plan are those of gemm256.hip (rows of 128 B, 16-byte chunk p of row r at p ^ ((r >> 1) & 7)).
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

using namespace gemm;

namespace {

constexpr int BM = 256, BN = 256;
constexpr int TILE = 256 * 128;      // 32 KiB per operand per K tile
constexpr int STAGE = 2 * TILE;      // activation rows, then weight rows
constexpr int LDS_BYTES = 2 * STAGE;  // 128 KiB

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define SLOT() __builtin_amdgcn_sched_barrier(0)

template <int... S, typename F>
__device__ __forceinline__ void for_each_slot(std::integer_sequence<int, S...>, F&& f) {
  (f(std::integral_constant<int, S>{}), ...);
}

// ---- the 64 accumulators (tile idx = 8 j + i -> a[4 idx : 4 idx + 3]) live in AGPRs that only these asm statements name: left to the
// register allocator the 256 accumulator registers fill the AGPR file and every loop back-edge costs ~100 single-register moves
// (v_accvgpr_mov / read / write) to permute them.  (Generated text: tools-free, 64 cases each.)
__device__ __forceinline__ void acc_mfma(int idx, bf16x8 wfrag, bf16x8 xfrag) {
  switch (idx) {
    case 0: asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], %0, %1, a[0:3]" ::"v"(wfrag), "v"(xfrag) : "a0", "a1", "a2", "a3"); break;
    case 1: asm volatile("v_mfma_f32_16x16x32_bf16 a[4:7], %0, %1, a[4:7]" ::"v"(wfrag), "v"(xfrag) : "a4", "a5", "a6", "a7"); break;
    case 2: asm volatile("v_mfma_f32_16x16x32_bf16 a[8:11], %0, %1, a[8:11]" ::"v"(wfrag), "v"(xfrag) : "a8", "a9", "a10", "a11"); break;
    case 3: asm volatile("v_mfma_f32_16x16x32_bf16 a[12:15], %0, %1, a[12:15]" ::"v"(wfrag), "v"(xfrag) : "a12", "a13", "a14", "a15"); break;
    case 4: asm volatile("v_mfma_f32_16x16x32_bf16 a[16:19], %0, %1, a[16:19]" ::"v"(wfrag), "v"(xfrag) : "a16", "a17", "a18", "a19"); break;
    case 5: asm volatile("v_mfma_f32_16x16x32_bf16 a[20:23], %0, %1, a[20:23]" ::"v"(wfrag), "v"(xfrag) : "a20", "a21", "a22", "a23"); break;
    case 6: asm volatile("v_mfma_f32_16x16x32_bf16 a[24:27], %0, %1, a[24:27]" ::"v"(wfrag), "v"(xfrag) : "a24", "a25", "a26", "a27"); break;
    case 7: asm volatile("v_mfma_f32_16x16x32_bf16 a[28:31], %0, %1, a[28:31]" ::"v"(wfrag), "v"(xfrag) : "a28", "a29", "a30", "a31"); break;
    case 8: asm volatile("v_mfma_f32_16x16x32_bf16 a[32:35], %0, %1, a[32:35]" ::"v"(wfrag), "v"(xfrag) : "a32", "a33", "a34", "a35"); break;
    case 9: asm volatile("v_mfma_f32_16x16x32_bf16 a[36:39], %0, %1, a[36:39]" ::"v"(wfrag), "v"(xfrag) : "a36", "a37", "a38", "a39"); break;
    case 10: asm volatile("v_mfma_f32_16x16x32_bf16 a[40:43], %0, %1, a[40:43]" ::"v"(wfrag), "v"(xfrag) : "a40", "a41", "a42", "a43"); break;
    case 11: asm volatile("v_mfma_f32_16x16x32_bf16 a[44:47], %0, %1, a[44:47]" ::"v"(wfrag), "v"(xfrag) : "a44", "a45", "a46", "a47"); break;
    case 12: asm volatile("v_mfma_f32_16x16x32_bf16 a[48:51], %0, %1, a[48:51]" ::"v"(wfrag), "v"(xfrag) : "a48", "a49", "a50", "a51"); break;
    case 13: asm volatile("v_mfma_f32_16x16x32_bf16 a[52:55], %0, %1, a[52:55]" ::"v"(wfrag), "v"(xfrag) : "a52", "a53", "a54", "a55"); break;
    case 14: asm volatile("v_mfma_f32_16x16x32_bf16 a[56:59], %0, %1, a[56:59]" ::"v"(wfrag), "v"(xfrag) : "a56", "a57", "a58", "a59"); break;
    case 15: asm volatile("v_mfma_f32_16x16x32_bf16 a[60:63], %0, %1, a[60:63]" ::"v"(wfrag), "v"(xfrag) : "a60", "a61", "a62", "a63"); break;
    case 16: asm volatile("v_mfma_f32_16x16x32_bf16 a[64:67], %0, %1, a[64:67]" ::"v"(wfrag), "v"(xfrag) : "a64", "a65", "a66", "a67"); break;
    case 17: asm volatile("v_mfma_f32_16x16x32_bf16 a[68:71], %0, %1, a[68:71]" ::"v"(wfrag), "v"(xfrag) : "a68", "a69", "a70", "a71"); break;
    case 18: asm volatile("v_mfma_f32_16x16x32_bf16 a[72:75], %0, %1, a[72:75]" ::"v"(wfrag), "v"(xfrag) : "a72", "a73", "a74", "a75"); break;
    case 19: asm volatile("v_mfma_f32_16x16x32_bf16 a[76:79], %0, %1, a[76:79]" ::"v"(wfrag), "v"(xfrag) : "a76", "a77", "a78", "a79"); break;
    case 20: asm volatile("v_mfma_f32_16x16x32_bf16 a[80:83], %0, %1, a[80:83]" ::"v"(wfrag), "v"(xfrag) : "a80", "a81", "a82", "a83"); break;
    case 21: asm volatile("v_mfma_f32_16x16x32_bf16 a[84:87], %0, %1, a[84:87]" ::"v"(wfrag), "v"(xfrag) : "a84", "a85", "a86", "a87"); break;
    case 22: asm volatile("v_mfma_f32_16x16x32_bf16 a[88:91], %0, %1, a[88:91]" ::"v"(wfrag), "v"(xfrag) : "a88", "a89", "a90", "a91"); break;
    case 23: asm volatile("v_mfma_f32_16x16x32_bf16 a[92:95], %0, %1, a[92:95]" ::"v"(wfrag), "v"(xfrag) : "a92", "a93", "a94", "a95"); break;
    case 24: asm volatile("v_mfma_f32_16x16x32_bf16 a[96:99], %0, %1, a[96:99]" ::"v"(wfrag), "v"(xfrag) : "a96", "a97", "a98", "a99"); break;
    case 25: asm volatile("v_mfma_f32_16x16x32_bf16 a[100:103], %0, %1, a[100:103]" ::"v"(wfrag), "v"(xfrag) : "a100", "a101", "a102", "a103"); break;
    case 26: asm volatile("v_mfma_f32_16x16x32_bf16 a[104:107], %0, %1, a[104:107]" ::"v"(wfrag), "v"(xfrag) : "a104", "a105", "a106", "a107"); break;
    case 27: asm volatile("v_mfma_f32_16x16x32_bf16 a[108:111], %0, %1, a[108:111]" ::"v"(wfrag), "v"(xfrag) : "a108", "a109", "a110", "a111"); break;
    case 28: asm volatile("v_mfma_f32_16x16x32_bf16 a[112:115], %0, %1, a[112:115]" ::"v"(wfrag), "v"(xfrag) : "a112", "a113", "a114", "a115"); break;
    case 29: asm volatile("v_mfma_f32_16x16x32_bf16 a[116:119], %0, %1, a[116:119]" ::"v"(wfrag), "v"(xfrag) : "a116", "a117", "a118", "a119"); break;
    case 30: asm volatile("v_mfma_f32_16x16x32_bf16 a[120:123], %0, %1, a[120:123]" ::"v"(wfrag), "v"(xfrag) : "a120", "a121", "a122", "a123"); break;
    case 31: asm volatile("v_mfma_f32_16x16x32_bf16 a[124:127], %0, %1, a[124:127]" ::"v"(wfrag), "v"(xfrag) : "a124", "a125", "a126", "a127"); break;
    case 32: asm volatile("v_mfma_f32_16x16x32_bf16 a[128:131], %0, %1, a[128:131]" ::"v"(wfrag), "v"(xfrag) : "a128", "a129", "a130", "a131"); break;
    case 33: asm volatile("v_mfma_f32_16x16x32_bf16 a[132:135], %0, %1, a[132:135]" ::"v"(wfrag), "v"(xfrag) : "a132", "a133", "a134", "a135"); break;
    case 34: asm volatile("v_mfma_f32_16x16x32_bf16 a[136:139], %0, %1, a[136:139]" ::"v"(wfrag), "v"(xfrag) : "a136", "a137", "a138", "a139"); break;
    case 35: asm volatile("v_mfma_f32_16x16x32_bf16 a[140:143], %0, %1, a[140:143]" ::"v"(wfrag), "v"(xfrag) : "a140", "a141", "a142", "a143"); break;
    case 36: asm volatile("v_mfma_f32_16x16x32_bf16 a[144:147], %0, %1, a[144:147]" ::"v"(wfrag), "v"(xfrag) : "a144", "a145", "a146", "a147"); break;
    case 37: asm volatile("v_mfma_f32_16x16x32_bf16 a[148:151], %0, %1, a[148:151]" ::"v"(wfrag), "v"(xfrag) : "a148", "a149", "a150", "a151"); break;
    case 38: asm volatile("v_mfma_f32_16x16x32_bf16 a[152:155], %0, %1, a[152:155]" ::"v"(wfrag), "v"(xfrag) : "a152", "a153", "a154", "a155"); break;
    case 39: asm volatile("v_mfma_f32_16x16x32_bf16 a[156:159], %0, %1, a[156:159]" ::"v"(wfrag), "v"(xfrag) : "a156", "a157", "a158", "a159"); break;
    case 40: asm volatile("v_mfma_f32_16x16x32_bf16 a[160:163], %0, %1, a[160:163]" ::"v"(wfrag), "v"(xfrag) : "a160", "a161", "a162", "a163"); break;
    case 41: asm volatile("v_mfma_f32_16x16x32_bf16 a[164:167], %0, %1, a[164:167]" ::"v"(wfrag), "v"(xfrag) : "a164", "a165", "a166", "a167"); break;
    case 42: asm volatile("v_mfma_f32_16x16x32_bf16 a[168:171], %0, %1, a[168:171]" ::"v"(wfrag), "v"(xfrag) : "a168", "a169", "a170", "a171"); break;
    case 43: asm volatile("v_mfma_f32_16x16x32_bf16 a[172:175], %0, %1, a[172:175]" ::"v"(wfrag), "v"(xfrag) : "a172", "a173", "a174", "a175"); break;
    case 44: asm volatile("v_mfma_f32_16x16x32_bf16 a[176:179], %0, %1, a[176:179]" ::"v"(wfrag), "v"(xfrag) : "a176", "a177", "a178", "a179"); break;
    case 45: asm volatile("v_mfma_f32_16x16x32_bf16 a[180:183], %0, %1, a[180:183]" ::"v"(wfrag), "v"(xfrag) : "a180", "a181", "a182", "a183"); break;
    case 46: asm volatile("v_mfma_f32_16x16x32_bf16 a[184:187], %0, %1, a[184:187]" ::"v"(wfrag), "v"(xfrag) : "a184", "a185", "a186", "a187"); break;
    case 47: asm volatile("v_mfma_f32_16x16x32_bf16 a[188:191], %0, %1, a[188:191]" ::"v"(wfrag), "v"(xfrag) : "a188", "a189", "a190", "a191"); break;
    case 48: asm volatile("v_mfma_f32_16x16x32_bf16 a[192:195], %0, %1, a[192:195]" ::"v"(wfrag), "v"(xfrag) : "a192", "a193", "a194", "a195"); break;
    case 49: asm volatile("v_mfma_f32_16x16x32_bf16 a[196:199], %0, %1, a[196:199]" ::"v"(wfrag), "v"(xfrag) : "a196", "a197", "a198", "a199"); break;
    case 50: asm volatile("v_mfma_f32_16x16x32_bf16 a[200:203], %0, %1, a[200:203]" ::"v"(wfrag), "v"(xfrag) : "a200", "a201", "a202", "a203"); break;
    case 51: asm volatile("v_mfma_f32_16x16x32_bf16 a[204:207], %0, %1, a[204:207]" ::"v"(wfrag), "v"(xfrag) : "a204", "a205", "a206", "a207"); break;
    case 52: asm volatile("v_mfma_f32_16x16x32_bf16 a[208:211], %0, %1, a[208:211]" ::"v"(wfrag), "v"(xfrag) : "a208", "a209", "a210", "a211"); break;
    case 53: asm volatile("v_mfma_f32_16x16x32_bf16 a[212:215], %0, %1, a[212:215]" ::"v"(wfrag), "v"(xfrag) : "a212", "a213", "a214", "a215"); break;
    case 54: asm volatile("v_mfma_f32_16x16x32_bf16 a[216:219], %0, %1, a[216:219]" ::"v"(wfrag), "v"(xfrag) : "a216", "a217", "a218", "a219"); break;
    case 55: asm volatile("v_mfma_f32_16x16x32_bf16 a[220:223], %0, %1, a[220:223]" ::"v"(wfrag), "v"(xfrag) : "a220", "a221", "a222", "a223"); break;
    case 56: asm volatile("v_mfma_f32_16x16x32_bf16 a[224:227], %0, %1, a[224:227]" ::"v"(wfrag), "v"(xfrag) : "a224", "a225", "a226", "a227"); break;
    case 57: asm volatile("v_mfma_f32_16x16x32_bf16 a[228:231], %0, %1, a[228:231]" ::"v"(wfrag), "v"(xfrag) : "a228", "a229", "a230", "a231"); break;
    case 58: asm volatile("v_mfma_f32_16x16x32_bf16 a[232:235], %0, %1, a[232:235]" ::"v"(wfrag), "v"(xfrag) : "a232", "a233", "a234", "a235"); break;
    case 59: asm volatile("v_mfma_f32_16x16x32_bf16 a[236:239], %0, %1, a[236:239]" ::"v"(wfrag), "v"(xfrag) : "a236", "a237", "a238", "a239"); break;
    case 60: asm volatile("v_mfma_f32_16x16x32_bf16 a[240:243], %0, %1, a[240:243]" ::"v"(wfrag), "v"(xfrag) : "a240", "a241", "a242", "a243"); break;
    case 61: asm volatile("v_mfma_f32_16x16x32_bf16 a[244:247], %0, %1, a[244:247]" ::"v"(wfrag), "v"(xfrag) : "a244", "a245", "a246", "a247"); break;
    case 62: asm volatile("v_mfma_f32_16x16x32_bf16 a[248:251], %0, %1, a[248:251]" ::"v"(wfrag), "v"(xfrag) : "a248", "a249", "a250", "a251"); break;
    case 63: asm volatile("v_mfma_f32_16x16x32_bf16 a[252:255], %0, %1, a[252:255]" ::"v"(wfrag), "v"(xfrag) : "a252", "a253", "a254", "a255"); break;
  }
}
__device__ __forceinline__ void acc_zero(int idx, float z) {
  switch (idx) {
    case 0: asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0" ::"v"(z) : "a0", "a1", "a2", "a3"); break;
    case 1: asm volatile("v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0" ::"v"(z) : "a4", "a5", "a6", "a7"); break;
    case 2: asm volatile("v_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %0\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_write_b32 a11, %0" ::"v"(z) : "a8", "a9", "a10", "a11"); break;
    case 3: asm volatile("v_accvgpr_write_b32 a12, %0\n\tv_accvgpr_write_b32 a13, %0\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_write_b32 a15, %0" ::"v"(z) : "a12", "a13", "a14", "a15"); break;
    case 4: asm volatile("v_accvgpr_write_b32 a16, %0\n\tv_accvgpr_write_b32 a17, %0\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_write_b32 a19, %0" ::"v"(z) : "a16", "a17", "a18", "a19"); break;
    case 5: asm volatile("v_accvgpr_write_b32 a20, %0\n\tv_accvgpr_write_b32 a21, %0\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_write_b32 a23, %0" ::"v"(z) : "a20", "a21", "a22", "a23"); break;
    case 6: asm volatile("v_accvgpr_write_b32 a24, %0\n\tv_accvgpr_write_b32 a25, %0\n\tv_accvgpr_write_b32 a26, %0\n\tv_accvgpr_write_b32 a27, %0" ::"v"(z) : "a24", "a25", "a26", "a27"); break;
    case 7: asm volatile("v_accvgpr_write_b32 a28, %0\n\tv_accvgpr_write_b32 a29, %0\n\tv_accvgpr_write_b32 a30, %0\n\tv_accvgpr_write_b32 a31, %0" ::"v"(z) : "a28", "a29", "a30", "a31"); break;
    case 8: asm volatile("v_accvgpr_write_b32 a32, %0\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_write_b32 a35, %0" ::"v"(z) : "a32", "a33", "a34", "a35"); break;
    case 9: asm volatile("v_accvgpr_write_b32 a36, %0\n\tv_accvgpr_write_b32 a37, %0\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_write_b32 a39, %0" ::"v"(z) : "a36", "a37", "a38", "a39"); break;
    case 10: asm volatile("v_accvgpr_write_b32 a40, %0\n\tv_accvgpr_write_b32 a41, %0\n\tv_accvgpr_write_b32 a42, %0\n\tv_accvgpr_write_b32 a43, %0" ::"v"(z) : "a40", "a41", "a42", "a43"); break;
    case 11: asm volatile("v_accvgpr_write_b32 a44, %0\n\tv_accvgpr_write_b32 a45, %0\n\tv_accvgpr_write_b32 a46, %0\n\tv_accvgpr_write_b32 a47, %0" ::"v"(z) : "a44", "a45", "a46", "a47"); break;
    case 12: asm volatile("v_accvgpr_write_b32 a48, %0\n\tv_accvgpr_write_b32 a49, %0\n\tv_accvgpr_write_b32 a50, %0\n\tv_accvgpr_write_b32 a51, %0" ::"v"(z) : "a48", "a49", "a50", "a51"); break;
    case 13: asm volatile("v_accvgpr_write_b32 a52, %0\n\tv_accvgpr_write_b32 a53, %0\n\tv_accvgpr_write_b32 a54, %0\n\tv_accvgpr_write_b32 a55, %0" ::"v"(z) : "a52", "a53", "a54", "a55"); break;
    case 14: asm volatile("v_accvgpr_write_b32 a56, %0\n\tv_accvgpr_write_b32 a57, %0\n\tv_accvgpr_write_b32 a58, %0\n\tv_accvgpr_write_b32 a59, %0" ::"v"(z) : "a56", "a57", "a58", "a59"); break;
    case 15: asm volatile("v_accvgpr_write_b32 a60, %0\n\tv_accvgpr_write_b32 a61, %0\n\tv_accvgpr_write_b32 a62, %0\n\tv_accvgpr_write_b32 a63, %0" ::"v"(z) : "a60", "a61", "a62", "a63"); break;
    case 16: asm volatile("v_accvgpr_write_b32 a64, %0\n\tv_accvgpr_write_b32 a65, %0\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_write_b32 a67, %0" ::"v"(z) : "a64", "a65", "a66", "a67"); break;
    case 17: asm volatile("v_accvgpr_write_b32 a68, %0\n\tv_accvgpr_write_b32 a69, %0\n\tv_accvgpr_write_b32 a70, %0\n\tv_accvgpr_write_b32 a71, %0" ::"v"(z) : "a68", "a69", "a70", "a71"); break;
    case 18: asm volatile("v_accvgpr_write_b32 a72, %0\n\tv_accvgpr_write_b32 a73, %0\n\tv_accvgpr_write_b32 a74, %0\n\tv_accvgpr_write_b32 a75, %0" ::"v"(z) : "a72", "a73", "a74", "a75"); break;
    case 19: asm volatile("v_accvgpr_write_b32 a76, %0\n\tv_accvgpr_write_b32 a77, %0\n\tv_accvgpr_write_b32 a78, %0\n\tv_accvgpr_write_b32 a79, %0" ::"v"(z) : "a76", "a77", "a78", "a79"); break;
    case 20: asm volatile("v_accvgpr_write_b32 a80, %0\n\tv_accvgpr_write_b32 a81, %0\n\tv_accvgpr_write_b32 a82, %0\n\tv_accvgpr_write_b32 a83, %0" ::"v"(z) : "a80", "a81", "a82", "a83"); break;
    case 21: asm volatile("v_accvgpr_write_b32 a84, %0\n\tv_accvgpr_write_b32 a85, %0\n\tv_accvgpr_write_b32 a86, %0\n\tv_accvgpr_write_b32 a87, %0" ::"v"(z) : "a84", "a85", "a86", "a87"); break;
    case 22: asm volatile("v_accvgpr_write_b32 a88, %0\n\tv_accvgpr_write_b32 a89, %0\n\tv_accvgpr_write_b32 a90, %0\n\tv_accvgpr_write_b32 a91, %0" ::"v"(z) : "a88", "a89", "a90", "a91"); break;
    case 23: asm volatile("v_accvgpr_write_b32 a92, %0\n\tv_accvgpr_write_b32 a93, %0\n\tv_accvgpr_write_b32 a94, %0\n\tv_accvgpr_write_b32 a95, %0" ::"v"(z) : "a92", "a93", "a94", "a95"); break;
    case 24: asm volatile("v_accvgpr_write_b32 a96, %0\n\tv_accvgpr_write_b32 a97, %0\n\tv_accvgpr_write_b32 a98, %0\n\tv_accvgpr_write_b32 a99, %0" ::"v"(z) : "a96", "a97", "a98", "a99"); break;
    case 25: asm volatile("v_accvgpr_write_b32 a100, %0\n\tv_accvgpr_write_b32 a101, %0\n\tv_accvgpr_write_b32 a102, %0\n\tv_accvgpr_write_b32 a103, %0" ::"v"(z) : "a100", "a101", "a102", "a103"); break;
    case 26: asm volatile("v_accvgpr_write_b32 a104, %0\n\tv_accvgpr_write_b32 a105, %0\n\tv_accvgpr_write_b32 a106, %0\n\tv_accvgpr_write_b32 a107, %0" ::"v"(z) : "a104", "a105", "a106", "a107"); break;
    case 27: asm volatile("v_accvgpr_write_b32 a108, %0\n\tv_accvgpr_write_b32 a109, %0\n\tv_accvgpr_write_b32 a110, %0\n\tv_accvgpr_write_b32 a111, %0" ::"v"(z) : "a108", "a109", "a110", "a111"); break;
    case 28: asm volatile("v_accvgpr_write_b32 a112, %0\n\tv_accvgpr_write_b32 a113, %0\n\tv_accvgpr_write_b32 a114, %0\n\tv_accvgpr_write_b32 a115, %0" ::"v"(z) : "a112", "a113", "a114", "a115"); break;
    case 29: asm volatile("v_accvgpr_write_b32 a116, %0\n\tv_accvgpr_write_b32 a117, %0\n\tv_accvgpr_write_b32 a118, %0\n\tv_accvgpr_write_b32 a119, %0" ::"v"(z) : "a116", "a117", "a118", "a119"); break;
    case 30: asm volatile("v_accvgpr_write_b32 a120, %0\n\tv_accvgpr_write_b32 a121, %0\n\tv_accvgpr_write_b32 a122, %0\n\tv_accvgpr_write_b32 a123, %0" ::"v"(z) : "a120", "a121", "a122", "a123"); break;
    case 31: asm volatile("v_accvgpr_write_b32 a124, %0\n\tv_accvgpr_write_b32 a125, %0\n\tv_accvgpr_write_b32 a126, %0\n\tv_accvgpr_write_b32 a127, %0" ::"v"(z) : "a124", "a125", "a126", "a127"); break;
    case 32: asm volatile("v_accvgpr_write_b32 a128, %0\n\tv_accvgpr_write_b32 a129, %0\n\tv_accvgpr_write_b32 a130, %0\n\tv_accvgpr_write_b32 a131, %0" ::"v"(z) : "a128", "a129", "a130", "a131"); break;
    case 33: asm volatile("v_accvgpr_write_b32 a132, %0\n\tv_accvgpr_write_b32 a133, %0\n\tv_accvgpr_write_b32 a134, %0\n\tv_accvgpr_write_b32 a135, %0" ::"v"(z) : "a132", "a133", "a134", "a135"); break;
    case 34: asm volatile("v_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %0\n\tv_accvgpr_write_b32 a138, %0\n\tv_accvgpr_write_b32 a139, %0" ::"v"(z) : "a136", "a137", "a138", "a139"); break;
    case 35: asm volatile("v_accvgpr_write_b32 a140, %0\n\tv_accvgpr_write_b32 a141, %0\n\tv_accvgpr_write_b32 a142, %0\n\tv_accvgpr_write_b32 a143, %0" ::"v"(z) : "a140", "a141", "a142", "a143"); break;
    case 36: asm volatile("v_accvgpr_write_b32 a144, %0\n\tv_accvgpr_write_b32 a145, %0\n\tv_accvgpr_write_b32 a146, %0\n\tv_accvgpr_write_b32 a147, %0" ::"v"(z) : "a144", "a145", "a146", "a147"); break;
    case 37: asm volatile("v_accvgpr_write_b32 a148, %0\n\tv_accvgpr_write_b32 a149, %0\n\tv_accvgpr_write_b32 a150, %0\n\tv_accvgpr_write_b32 a151, %0" ::"v"(z) : "a148", "a149", "a150", "a151"); break;
    case 38: asm volatile("v_accvgpr_write_b32 a152, %0\n\tv_accvgpr_write_b32 a153, %0\n\tv_accvgpr_write_b32 a154, %0\n\tv_accvgpr_write_b32 a155, %0" ::"v"(z) : "a152", "a153", "a154", "a155"); break;
    case 39: asm volatile("v_accvgpr_write_b32 a156, %0\n\tv_accvgpr_write_b32 a157, %0\n\tv_accvgpr_write_b32 a158, %0\n\tv_accvgpr_write_b32 a159, %0" ::"v"(z) : "a156", "a157", "a158", "a159"); break;
    case 40: asm volatile("v_accvgpr_write_b32 a160, %0\n\tv_accvgpr_write_b32 a161, %0\n\tv_accvgpr_write_b32 a162, %0\n\tv_accvgpr_write_b32 a163, %0" ::"v"(z) : "a160", "a161", "a162", "a163"); break;
    case 41: asm volatile("v_accvgpr_write_b32 a164, %0\n\tv_accvgpr_write_b32 a165, %0\n\tv_accvgpr_write_b32 a166, %0\n\tv_accvgpr_write_b32 a167, %0" ::"v"(z) : "a164", "a165", "a166", "a167"); break;
    case 42: asm volatile("v_accvgpr_write_b32 a168, %0\n\tv_accvgpr_write_b32 a169, %0\n\tv_accvgpr_write_b32 a170, %0\n\tv_accvgpr_write_b32 a171, %0" ::"v"(z) : "a168", "a169", "a170", "a171"); break;
    case 43: asm volatile("v_accvgpr_write_b32 a172, %0\n\tv_accvgpr_write_b32 a173, %0\n\tv_accvgpr_write_b32 a174, %0\n\tv_accvgpr_write_b32 a175, %0" ::"v"(z) : "a172", "a173", "a174", "a175"); break;
    case 44: asm volatile("v_accvgpr_write_b32 a176, %0\n\tv_accvgpr_write_b32 a177, %0\n\tv_accvgpr_write_b32 a178, %0\n\tv_accvgpr_write_b32 a179, %0" ::"v"(z) : "a176", "a177", "a178", "a179"); break;
    case 45: asm volatile("v_accvgpr_write_b32 a180, %0\n\tv_accvgpr_write_b32 a181, %0\n\tv_accvgpr_write_b32 a182, %0\n\tv_accvgpr_write_b32 a183, %0" ::"v"(z) : "a180", "a181", "a182", "a183"); break;
    case 46: asm volatile("v_accvgpr_write_b32 a184, %0\n\tv_accvgpr_write_b32 a185, %0\n\tv_accvgpr_write_b32 a186, %0\n\tv_accvgpr_write_b32 a187, %0" ::"v"(z) : "a184", "a185", "a186", "a187"); break;
    case 47: asm volatile("v_accvgpr_write_b32 a188, %0\n\tv_accvgpr_write_b32 a189, %0\n\tv_accvgpr_write_b32 a190, %0\n\tv_accvgpr_write_b32 a191, %0" ::"v"(z) : "a188", "a189", "a190", "a191"); break;
    case 48: asm volatile("v_accvgpr_write_b32 a192, %0\n\tv_accvgpr_write_b32 a193, %0\n\tv_accvgpr_write_b32 a194, %0\n\tv_accvgpr_write_b32 a195, %0" ::"v"(z) : "a192", "a193", "a194", "a195"); break;
    case 49: asm volatile("v_accvgpr_write_b32 a196, %0\n\tv_accvgpr_write_b32 a197, %0\n\tv_accvgpr_write_b32 a198, %0\n\tv_accvgpr_write_b32 a199, %0" ::"v"(z) : "a196", "a197", "a198", "a199"); break;
    case 50: asm volatile("v_accvgpr_write_b32 a200, %0\n\tv_accvgpr_write_b32 a201, %0\n\tv_accvgpr_write_b32 a202, %0\n\tv_accvgpr_write_b32 a203, %0" ::"v"(z) : "a200", "a201", "a202", "a203"); break;
    case 51: asm volatile("v_accvgpr_write_b32 a204, %0\n\tv_accvgpr_write_b32 a205, %0\n\tv_accvgpr_write_b32 a206, %0\n\tv_accvgpr_write_b32 a207, %0" ::"v"(z) : "a204", "a205", "a206", "a207"); break;
    case 52: asm volatile("v_accvgpr_write_b32 a208, %0\n\tv_accvgpr_write_b32 a209, %0\n\tv_accvgpr_write_b32 a210, %0\n\tv_accvgpr_write_b32 a211, %0" ::"v"(z) : "a208", "a209", "a210", "a211"); break;
    case 53: asm volatile("v_accvgpr_write_b32 a212, %0\n\tv_accvgpr_write_b32 a213, %0\n\tv_accvgpr_write_b32 a214, %0\n\tv_accvgpr_write_b32 a215, %0" ::"v"(z) : "a212", "a213", "a214", "a215"); break;
    case 54: asm volatile("v_accvgpr_write_b32 a216, %0\n\tv_accvgpr_write_b32 a217, %0\n\tv_accvgpr_write_b32 a218, %0\n\tv_accvgpr_write_b32 a219, %0" ::"v"(z) : "a216", "a217", "a218", "a219"); break;
    case 55: asm volatile("v_accvgpr_write_b32 a220, %0\n\tv_accvgpr_write_b32 a221, %0\n\tv_accvgpr_write_b32 a222, %0\n\tv_accvgpr_write_b32 a223, %0" ::"v"(z) : "a220", "a221", "a222", "a223"); break;
    case 56: asm volatile("v_accvgpr_write_b32 a224, %0\n\tv_accvgpr_write_b32 a225, %0\n\tv_accvgpr_write_b32 a226, %0\n\tv_accvgpr_write_b32 a227, %0" ::"v"(z) : "a224", "a225", "a226", "a227"); break;
    case 57: asm volatile("v_accvgpr_write_b32 a228, %0\n\tv_accvgpr_write_b32 a229, %0\n\tv_accvgpr_write_b32 a230, %0\n\tv_accvgpr_write_b32 a231, %0" ::"v"(z) : "a228", "a229", "a230", "a231"); break;
    case 58: asm volatile("v_accvgpr_write_b32 a232, %0\n\tv_accvgpr_write_b32 a233, %0\n\tv_accvgpr_write_b32 a234, %0\n\tv_accvgpr_write_b32 a235, %0" ::"v"(z) : "a232", "a233", "a234", "a235"); break;
    case 59: asm volatile("v_accvgpr_write_b32 a236, %0\n\tv_accvgpr_write_b32 a237, %0\n\tv_accvgpr_write_b32 a238, %0\n\tv_accvgpr_write_b32 a239, %0" ::"v"(z) : "a236", "a237", "a238", "a239"); break;
    case 60: asm volatile("v_accvgpr_write_b32 a240, %0\n\tv_accvgpr_write_b32 a241, %0\n\tv_accvgpr_write_b32 a242, %0\n\tv_accvgpr_write_b32 a243, %0" ::"v"(z) : "a240", "a241", "a242", "a243"); break;
    case 61: asm volatile("v_accvgpr_write_b32 a244, %0\n\tv_accvgpr_write_b32 a245, %0\n\tv_accvgpr_write_b32 a246, %0\n\tv_accvgpr_write_b32 a247, %0" ::"v"(z) : "a244", "a245", "a246", "a247"); break;
    case 62: asm volatile("v_accvgpr_write_b32 a248, %0\n\tv_accvgpr_write_b32 a249, %0\n\tv_accvgpr_write_b32 a250, %0\n\tv_accvgpr_write_b32 a251, %0" ::"v"(z) : "a248", "a249", "a250", "a251"); break;
    case 63: asm volatile("v_accvgpr_write_b32 a252, %0\n\tv_accvgpr_write_b32 a253, %0\n\tv_accvgpr_write_b32 a254, %0\n\tv_accvgpr_write_b32 a255, %0" ::"v"(z) : "a252", "a253", "a254", "a255"); break;
  }
}
__device__ __forceinline__ f32x4 acc_read(int idx) {
  float f0, f1, f2, f3;
  switch (idx) {
    default: case 0: asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a0", "a1", "a2", "a3"); break;
    case 1: asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a4", "a5", "a6", "a7"); break;
    case 2: asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_accvgpr_read_b32 %2, a10\n\tv_accvgpr_read_b32 %3, a11" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a8", "a9", "a10", "a11"); break;
    case 3: asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\tv_accvgpr_read_b32 %2, a14\n\tv_accvgpr_read_b32 %3, a15" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a12", "a13", "a14", "a15"); break;
    case 4: asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a16", "a17", "a18", "a19"); break;
    case 5: asm volatile("v_accvgpr_read_b32 %0, a20\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a22\n\tv_accvgpr_read_b32 %3, a23" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a20", "a21", "a22", "a23"); break;
    case 6: asm volatile("v_accvgpr_read_b32 %0, a24\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a26\n\tv_accvgpr_read_b32 %3, a27" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a24", "a25", "a26", "a27"); break;
    case 7: asm volatile("v_accvgpr_read_b32 %0, a28\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a30\n\tv_accvgpr_read_b32 %3, a31" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a28", "a29", "a30", "a31"); break;
    case 8: asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a32", "a33", "a34", "a35"); break;
    case 9: asm volatile("v_accvgpr_read_b32 %0, a36\n\tv_accvgpr_read_b32 %1, a37\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a39" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a36", "a37", "a38", "a39"); break;
    case 10: asm volatile("v_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a43" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a40", "a41", "a42", "a43"); break;
    case 11: asm volatile("v_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a47" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a44", "a45", "a46", "a47"); break;
    case 12: asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a48", "a49", "a50", "a51"); break;
    case 13: asm volatile("v_accvgpr_read_b32 %0, a52\n\tv_accvgpr_read_b32 %1, a53\n\tv_accvgpr_read_b32 %2, a54\n\tv_accvgpr_read_b32 %3, a55" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a52", "a53", "a54", "a55"); break;
    case 14: asm volatile("v_accvgpr_read_b32 %0, a56\n\tv_accvgpr_read_b32 %1, a57\n\tv_accvgpr_read_b32 %2, a58\n\tv_accvgpr_read_b32 %3, a59" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a56", "a57", "a58", "a59"); break;
    case 15: asm volatile("v_accvgpr_read_b32 %0, a60\n\tv_accvgpr_read_b32 %1, a61\n\tv_accvgpr_read_b32 %2, a62\n\tv_accvgpr_read_b32 %3, a63" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a60", "a61", "a62", "a63"); break;
    case 16: asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a64", "a65", "a66", "a67"); break;
    case 17: asm volatile("v_accvgpr_read_b32 %0, a68\n\tv_accvgpr_read_b32 %1, a69\n\tv_accvgpr_read_b32 %2, a70\n\tv_accvgpr_read_b32 %3, a71" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a68", "a69", "a70", "a71"); break;
    case 18: asm volatile("v_accvgpr_read_b32 %0, a72\n\tv_accvgpr_read_b32 %1, a73\n\tv_accvgpr_read_b32 %2, a74\n\tv_accvgpr_read_b32 %3, a75" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a72", "a73", "a74", "a75"); break;
    case 19: asm volatile("v_accvgpr_read_b32 %0, a76\n\tv_accvgpr_read_b32 %1, a77\n\tv_accvgpr_read_b32 %2, a78\n\tv_accvgpr_read_b32 %3, a79" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a76", "a77", "a78", "a79"); break;
    case 20: asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a80", "a81", "a82", "a83"); break;
    case 21: asm volatile("v_accvgpr_read_b32 %0, a84\n\tv_accvgpr_read_b32 %1, a85\n\tv_accvgpr_read_b32 %2, a86\n\tv_accvgpr_read_b32 %3, a87" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a84", "a85", "a86", "a87"); break;
    case 22: asm volatile("v_accvgpr_read_b32 %0, a88\n\tv_accvgpr_read_b32 %1, a89\n\tv_accvgpr_read_b32 %2, a90\n\tv_accvgpr_read_b32 %3, a91" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a88", "a89", "a90", "a91"); break;
    case 23: asm volatile("v_accvgpr_read_b32 %0, a92\n\tv_accvgpr_read_b32 %1, a93\n\tv_accvgpr_read_b32 %2, a94\n\tv_accvgpr_read_b32 %3, a95" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a92", "a93", "a94", "a95"); break;
    case 24: asm volatile("v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\tv_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a96", "a97", "a98", "a99"); break;
    case 25: asm volatile("v_accvgpr_read_b32 %0, a100\n\tv_accvgpr_read_b32 %1, a101\n\tv_accvgpr_read_b32 %2, a102\n\tv_accvgpr_read_b32 %3, a103" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a100", "a101", "a102", "a103"); break;
    case 26: asm volatile("v_accvgpr_read_b32 %0, a104\n\tv_accvgpr_read_b32 %1, a105\n\tv_accvgpr_read_b32 %2, a106\n\tv_accvgpr_read_b32 %3, a107" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a104", "a105", "a106", "a107"); break;
    case 27: asm volatile("v_accvgpr_read_b32 %0, a108\n\tv_accvgpr_read_b32 %1, a109\n\tv_accvgpr_read_b32 %2, a110\n\tv_accvgpr_read_b32 %3, a111" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a108", "a109", "a110", "a111"); break;
    case 28: asm volatile("v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\tv_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a112", "a113", "a114", "a115"); break;
    case 29: asm volatile("v_accvgpr_read_b32 %0, a116\n\tv_accvgpr_read_b32 %1, a117\n\tv_accvgpr_read_b32 %2, a118\n\tv_accvgpr_read_b32 %3, a119" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a116", "a117", "a118", "a119"); break;
    case 30: asm volatile("v_accvgpr_read_b32 %0, a120\n\tv_accvgpr_read_b32 %1, a121\n\tv_accvgpr_read_b32 %2, a122\n\tv_accvgpr_read_b32 %3, a123" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a120", "a121", "a122", "a123"); break;
    case 31: asm volatile("v_accvgpr_read_b32 %0, a124\n\tv_accvgpr_read_b32 %1, a125\n\tv_accvgpr_read_b32 %2, a126\n\tv_accvgpr_read_b32 %3, a127" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a124", "a125", "a126", "a127"); break;
    case 32: asm volatile("v_accvgpr_read_b32 %0, a128\n\tv_accvgpr_read_b32 %1, a129\n\tv_accvgpr_read_b32 %2, a130\n\tv_accvgpr_read_b32 %3, a131" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a128", "a129", "a130", "a131"); break;
    case 33: asm volatile("v_accvgpr_read_b32 %0, a132\n\tv_accvgpr_read_b32 %1, a133\n\tv_accvgpr_read_b32 %2, a134\n\tv_accvgpr_read_b32 %3, a135" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a132", "a133", "a134", "a135"); break;
    case 34: asm volatile("v_accvgpr_read_b32 %0, a136\n\tv_accvgpr_read_b32 %1, a137\n\tv_accvgpr_read_b32 %2, a138\n\tv_accvgpr_read_b32 %3, a139" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a136", "a137", "a138", "a139"); break;
    case 35: asm volatile("v_accvgpr_read_b32 %0, a140\n\tv_accvgpr_read_b32 %1, a141\n\tv_accvgpr_read_b32 %2, a142\n\tv_accvgpr_read_b32 %3, a143" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a140", "a141", "a142", "a143"); break;
    case 36: asm volatile("v_accvgpr_read_b32 %0, a144\n\tv_accvgpr_read_b32 %1, a145\n\tv_accvgpr_read_b32 %2, a146\n\tv_accvgpr_read_b32 %3, a147" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a144", "a145", "a146", "a147"); break;
    case 37: asm volatile("v_accvgpr_read_b32 %0, a148\n\tv_accvgpr_read_b32 %1, a149\n\tv_accvgpr_read_b32 %2, a150\n\tv_accvgpr_read_b32 %3, a151" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a148", "a149", "a150", "a151"); break;
    case 38: asm volatile("v_accvgpr_read_b32 %0, a152\n\tv_accvgpr_read_b32 %1, a153\n\tv_accvgpr_read_b32 %2, a154\n\tv_accvgpr_read_b32 %3, a155" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a152", "a153", "a154", "a155"); break;
    case 39: asm volatile("v_accvgpr_read_b32 %0, a156\n\tv_accvgpr_read_b32 %1, a157\n\tv_accvgpr_read_b32 %2, a158\n\tv_accvgpr_read_b32 %3, a159" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a156", "a157", "a158", "a159"); break;
    case 40: asm volatile("v_accvgpr_read_b32 %0, a160\n\tv_accvgpr_read_b32 %1, a161\n\tv_accvgpr_read_b32 %2, a162\n\tv_accvgpr_read_b32 %3, a163" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a160", "a161", "a162", "a163"); break;
    case 41: asm volatile("v_accvgpr_read_b32 %0, a164\n\tv_accvgpr_read_b32 %1, a165\n\tv_accvgpr_read_b32 %2, a166\n\tv_accvgpr_read_b32 %3, a167" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a164", "a165", "a166", "a167"); break;
    case 42: asm volatile("v_accvgpr_read_b32 %0, a168\n\tv_accvgpr_read_b32 %1, a169\n\tv_accvgpr_read_b32 %2, a170\n\tv_accvgpr_read_b32 %3, a171" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a168", "a169", "a170", "a171"); break;
    case 43: asm volatile("v_accvgpr_read_b32 %0, a172\n\tv_accvgpr_read_b32 %1, a173\n\tv_accvgpr_read_b32 %2, a174\n\tv_accvgpr_read_b32 %3, a175" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a172", "a173", "a174", "a175"); break;
    case 44: asm volatile("v_accvgpr_read_b32 %0, a176\n\tv_accvgpr_read_b32 %1, a177\n\tv_accvgpr_read_b32 %2, a178\n\tv_accvgpr_read_b32 %3, a179" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a176", "a177", "a178", "a179"); break;
    case 45: asm volatile("v_accvgpr_read_b32 %0, a180\n\tv_accvgpr_read_b32 %1, a181\n\tv_accvgpr_read_b32 %2, a182\n\tv_accvgpr_read_b32 %3, a183" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a180", "a181", "a182", "a183"); break;
    case 46: asm volatile("v_accvgpr_read_b32 %0, a184\n\tv_accvgpr_read_b32 %1, a185\n\tv_accvgpr_read_b32 %2, a186\n\tv_accvgpr_read_b32 %3, a187" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a184", "a185", "a186", "a187"); break;
    case 47: asm volatile("v_accvgpr_read_b32 %0, a188\n\tv_accvgpr_read_b32 %1, a189\n\tv_accvgpr_read_b32 %2, a190\n\tv_accvgpr_read_b32 %3, a191" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a188", "a189", "a190", "a191"); break;
    case 48: asm volatile("v_accvgpr_read_b32 %0, a192\n\tv_accvgpr_read_b32 %1, a193\n\tv_accvgpr_read_b32 %2, a194\n\tv_accvgpr_read_b32 %3, a195" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a192", "a193", "a194", "a195"); break;
    case 49: asm volatile("v_accvgpr_read_b32 %0, a196\n\tv_accvgpr_read_b32 %1, a197\n\tv_accvgpr_read_b32 %2, a198\n\tv_accvgpr_read_b32 %3, a199" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a196", "a197", "a198", "a199"); break;
    case 50: asm volatile("v_accvgpr_read_b32 %0, a200\n\tv_accvgpr_read_b32 %1, a201\n\tv_accvgpr_read_b32 %2, a202\n\tv_accvgpr_read_b32 %3, a203" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a200", "a201", "a202", "a203"); break;
    case 51: asm volatile("v_accvgpr_read_b32 %0, a204\n\tv_accvgpr_read_b32 %1, a205\n\tv_accvgpr_read_b32 %2, a206\n\tv_accvgpr_read_b32 %3, a207" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a204", "a205", "a206", "a207"); break;
    case 52: asm volatile("v_accvgpr_read_b32 %0, a208\n\tv_accvgpr_read_b32 %1, a209\n\tv_accvgpr_read_b32 %2, a210\n\tv_accvgpr_read_b32 %3, a211" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a208", "a209", "a210", "a211"); break;
    case 53: asm volatile("v_accvgpr_read_b32 %0, a212\n\tv_accvgpr_read_b32 %1, a213\n\tv_accvgpr_read_b32 %2, a214\n\tv_accvgpr_read_b32 %3, a215" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a212", "a213", "a214", "a215"); break;
    case 54: asm volatile("v_accvgpr_read_b32 %0, a216\n\tv_accvgpr_read_b32 %1, a217\n\tv_accvgpr_read_b32 %2, a218\n\tv_accvgpr_read_b32 %3, a219" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a216", "a217", "a218", "a219"); break;
    case 55: asm volatile("v_accvgpr_read_b32 %0, a220\n\tv_accvgpr_read_b32 %1, a221\n\tv_accvgpr_read_b32 %2, a222\n\tv_accvgpr_read_b32 %3, a223" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a220", "a221", "a222", "a223"); break;
    case 56: asm volatile("v_accvgpr_read_b32 %0, a224\n\tv_accvgpr_read_b32 %1, a225\n\tv_accvgpr_read_b32 %2, a226\n\tv_accvgpr_read_b32 %3, a227" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a224", "a225", "a226", "a227"); break;
    case 57: asm volatile("v_accvgpr_read_b32 %0, a228\n\tv_accvgpr_read_b32 %1, a229\n\tv_accvgpr_read_b32 %2, a230\n\tv_accvgpr_read_b32 %3, a231" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a228", "a229", "a230", "a231"); break;
    case 58: asm volatile("v_accvgpr_read_b32 %0, a232\n\tv_accvgpr_read_b32 %1, a233\n\tv_accvgpr_read_b32 %2, a234\n\tv_accvgpr_read_b32 %3, a235" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a232", "a233", "a234", "a235"); break;
    case 59: asm volatile("v_accvgpr_read_b32 %0, a236\n\tv_accvgpr_read_b32 %1, a237\n\tv_accvgpr_read_b32 %2, a238\n\tv_accvgpr_read_b32 %3, a239" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a236", "a237", "a238", "a239"); break;
    case 60: asm volatile("v_accvgpr_read_b32 %0, a240\n\tv_accvgpr_read_b32 %1, a241\n\tv_accvgpr_read_b32 %2, a242\n\tv_accvgpr_read_b32 %3, a243" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a240", "a241", "a242", "a243"); break;
    case 61: asm volatile("v_accvgpr_read_b32 %0, a244\n\tv_accvgpr_read_b32 %1, a245\n\tv_accvgpr_read_b32 %2, a246\n\tv_accvgpr_read_b32 %3, a247" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a244", "a245", "a246", "a247"); break;
    case 62: asm volatile("v_accvgpr_read_b32 %0, a248\n\tv_accvgpr_read_b32 %1, a249\n\tv_accvgpr_read_b32 %2, a250\n\tv_accvgpr_read_b32 %3, a251" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a248", "a249", "a250", "a251"); break;
    case 63: asm volatile("v_accvgpr_read_b32 %0, a252\n\tv_accvgpr_read_b32 %1, a253\n\tv_accvgpr_read_b32 %2, a254\n\tv_accvgpr_read_b32 %3, a255" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : : "a252", "a253", "a254", "a255"); break;
  }
  return f32x4{f0, f1, f2, f3};
}

// the first MFMA into an accumulator of an output tile takes C = 0 (no zeroing pass over 256 registers per tile)
__device__ __forceinline__ void acc_mfma_first(int idx, bf16x8 wfrag, bf16x8 xfrag) {
  switch (idx) {
    case 0: asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a0", "a1", "a2", "a3"); break;
    case 1: asm volatile("v_mfma_f32_16x16x32_bf16 a[4:7], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a4", "a5", "a6", "a7"); break;
    case 2: asm volatile("v_mfma_f32_16x16x32_bf16 a[8:11], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a8", "a9", "a10", "a11"); break;
    case 3: asm volatile("v_mfma_f32_16x16x32_bf16 a[12:15], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a12", "a13", "a14", "a15"); break;
    case 4: asm volatile("v_mfma_f32_16x16x32_bf16 a[16:19], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a16", "a17", "a18", "a19"); break;
    case 5: asm volatile("v_mfma_f32_16x16x32_bf16 a[20:23], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a20", "a21", "a22", "a23"); break;
    case 6: asm volatile("v_mfma_f32_16x16x32_bf16 a[24:27], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a24", "a25", "a26", "a27"); break;
    case 7: asm volatile("v_mfma_f32_16x16x32_bf16 a[28:31], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a28", "a29", "a30", "a31"); break;
    case 8: asm volatile("v_mfma_f32_16x16x32_bf16 a[32:35], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a32", "a33", "a34", "a35"); break;
    case 9: asm volatile("v_mfma_f32_16x16x32_bf16 a[36:39], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a36", "a37", "a38", "a39"); break;
    case 10: asm volatile("v_mfma_f32_16x16x32_bf16 a[40:43], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a40", "a41", "a42", "a43"); break;
    case 11: asm volatile("v_mfma_f32_16x16x32_bf16 a[44:47], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a44", "a45", "a46", "a47"); break;
    case 12: asm volatile("v_mfma_f32_16x16x32_bf16 a[48:51], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a48", "a49", "a50", "a51"); break;
    case 13: asm volatile("v_mfma_f32_16x16x32_bf16 a[52:55], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a52", "a53", "a54", "a55"); break;
    case 14: asm volatile("v_mfma_f32_16x16x32_bf16 a[56:59], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a56", "a57", "a58", "a59"); break;
    case 15: asm volatile("v_mfma_f32_16x16x32_bf16 a[60:63], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a60", "a61", "a62", "a63"); break;
    case 16: asm volatile("v_mfma_f32_16x16x32_bf16 a[64:67], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a64", "a65", "a66", "a67"); break;
    case 17: asm volatile("v_mfma_f32_16x16x32_bf16 a[68:71], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a68", "a69", "a70", "a71"); break;
    case 18: asm volatile("v_mfma_f32_16x16x32_bf16 a[72:75], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a72", "a73", "a74", "a75"); break;
    case 19: asm volatile("v_mfma_f32_16x16x32_bf16 a[76:79], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a76", "a77", "a78", "a79"); break;
    case 20: asm volatile("v_mfma_f32_16x16x32_bf16 a[80:83], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a80", "a81", "a82", "a83"); break;
    case 21: asm volatile("v_mfma_f32_16x16x32_bf16 a[84:87], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a84", "a85", "a86", "a87"); break;
    case 22: asm volatile("v_mfma_f32_16x16x32_bf16 a[88:91], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a88", "a89", "a90", "a91"); break;
    case 23: asm volatile("v_mfma_f32_16x16x32_bf16 a[92:95], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a92", "a93", "a94", "a95"); break;
    case 24: asm volatile("v_mfma_f32_16x16x32_bf16 a[96:99], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a96", "a97", "a98", "a99"); break;
    case 25: asm volatile("v_mfma_f32_16x16x32_bf16 a[100:103], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a100", "a101", "a102", "a103"); break;
    case 26: asm volatile("v_mfma_f32_16x16x32_bf16 a[104:107], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a104", "a105", "a106", "a107"); break;
    case 27: asm volatile("v_mfma_f32_16x16x32_bf16 a[108:111], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a108", "a109", "a110", "a111"); break;
    case 28: asm volatile("v_mfma_f32_16x16x32_bf16 a[112:115], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a112", "a113", "a114", "a115"); break;
    case 29: asm volatile("v_mfma_f32_16x16x32_bf16 a[116:119], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a116", "a117", "a118", "a119"); break;
    case 30: asm volatile("v_mfma_f32_16x16x32_bf16 a[120:123], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a120", "a121", "a122", "a123"); break;
    case 31: asm volatile("v_mfma_f32_16x16x32_bf16 a[124:127], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a124", "a125", "a126", "a127"); break;
    case 32: asm volatile("v_mfma_f32_16x16x32_bf16 a[128:131], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a128", "a129", "a130", "a131"); break;
    case 33: asm volatile("v_mfma_f32_16x16x32_bf16 a[132:135], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a132", "a133", "a134", "a135"); break;
    case 34: asm volatile("v_mfma_f32_16x16x32_bf16 a[136:139], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a136", "a137", "a138", "a139"); break;
    case 35: asm volatile("v_mfma_f32_16x16x32_bf16 a[140:143], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a140", "a141", "a142", "a143"); break;
    case 36: asm volatile("v_mfma_f32_16x16x32_bf16 a[144:147], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a144", "a145", "a146", "a147"); break;
    case 37: asm volatile("v_mfma_f32_16x16x32_bf16 a[148:151], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a148", "a149", "a150", "a151"); break;
    case 38: asm volatile("v_mfma_f32_16x16x32_bf16 a[152:155], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a152", "a153", "a154", "a155"); break;
    case 39: asm volatile("v_mfma_f32_16x16x32_bf16 a[156:159], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a156", "a157", "a158", "a159"); break;
    case 40: asm volatile("v_mfma_f32_16x16x32_bf16 a[160:163], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a160", "a161", "a162", "a163"); break;
    case 41: asm volatile("v_mfma_f32_16x16x32_bf16 a[164:167], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a164", "a165", "a166", "a167"); break;
    case 42: asm volatile("v_mfma_f32_16x16x32_bf16 a[168:171], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a168", "a169", "a170", "a171"); break;
    case 43: asm volatile("v_mfma_f32_16x16x32_bf16 a[172:175], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a172", "a173", "a174", "a175"); break;
    case 44: asm volatile("v_mfma_f32_16x16x32_bf16 a[176:179], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a176", "a177", "a178", "a179"); break;
    case 45: asm volatile("v_mfma_f32_16x16x32_bf16 a[180:183], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a180", "a181", "a182", "a183"); break;
    case 46: asm volatile("v_mfma_f32_16x16x32_bf16 a[184:187], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a184", "a185", "a186", "a187"); break;
    case 47: asm volatile("v_mfma_f32_16x16x32_bf16 a[188:191], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a188", "a189", "a190", "a191"); break;
    case 48: asm volatile("v_mfma_f32_16x16x32_bf16 a[192:195], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a192", "a193", "a194", "a195"); break;
    case 49: asm volatile("v_mfma_f32_16x16x32_bf16 a[196:199], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a196", "a197", "a198", "a199"); break;
    case 50: asm volatile("v_mfma_f32_16x16x32_bf16 a[200:203], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a200", "a201", "a202", "a203"); break;
    case 51: asm volatile("v_mfma_f32_16x16x32_bf16 a[204:207], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a204", "a205", "a206", "a207"); break;
    case 52: asm volatile("v_mfma_f32_16x16x32_bf16 a[208:211], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a208", "a209", "a210", "a211"); break;
    case 53: asm volatile("v_mfma_f32_16x16x32_bf16 a[212:215], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a212", "a213", "a214", "a215"); break;
    case 54: asm volatile("v_mfma_f32_16x16x32_bf16 a[216:219], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a216", "a217", "a218", "a219"); break;
    case 55: asm volatile("v_mfma_f32_16x16x32_bf16 a[220:223], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a220", "a221", "a222", "a223"); break;
    case 56: asm volatile("v_mfma_f32_16x16x32_bf16 a[224:227], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a224", "a225", "a226", "a227"); break;
    case 57: asm volatile("v_mfma_f32_16x16x32_bf16 a[228:231], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a228", "a229", "a230", "a231"); break;
    case 58: asm volatile("v_mfma_f32_16x16x32_bf16 a[232:235], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a232", "a233", "a234", "a235"); break;
    case 59: asm volatile("v_mfma_f32_16x16x32_bf16 a[236:239], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a236", "a237", "a238", "a239"); break;
    case 60: asm volatile("v_mfma_f32_16x16x32_bf16 a[240:243], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a240", "a241", "a242", "a243"); break;
    case 61: asm volatile("v_mfma_f32_16x16x32_bf16 a[244:247], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a244", "a245", "a246", "a247"); break;
    case 62: asm volatile("v_mfma_f32_16x16x32_bf16 a[248:251], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a248", "a249", "a250", "a251"); break;
    case 63: asm volatile("v_mfma_f32_16x16x32_bf16 a[252:255], %0, %1, 0" ::"v"(wfrag), "v"(xfrag) : "a252", "a253", "a254", "a255"); break;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The product form of the four-wave loop (taken by hwocr_gemm_wide256_w4's rule below; HWOCR_GEMM256=4: wherever it exists, =2: never):
// persistent tile loop as in gemm256.hip - the next tile's 32 prologue DMAs go out before the finished tile's
// epilogue - with the staged epilogues of gemm256.hip (bias / residual / activations through a 4-KiB-per-wave LDS region, whole
// 128-byte rows to HBM) walked over the wave's 128 x 128 block as two 64-column halves, and the gated ones (store_glu).
// bf16 only; the fused vision-QKV epilogue stays with the eight-wave kernel.
constexpr int W4_LDS = 2 * STAGE + 4 * 8192;  // the whole 160 KiB: two operand stages + 2 x 4 KiB of epilogue staging per wave

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_wide256w4_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
  const int ntiles = a.tilesM * a.tilesN;
  const int nk = a.K >> 6;  // >= 2 (launcher)
  int tile = blockIdx.x;
  int m0, n0;
  auto origin_of = [&](int id) {
    int tm, tn;
    tile_of_id(id, a.tilesM, a.tilesN, 4, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
  };
  const int r8 = lane >> 3, p = lane & 7;
  int voff[16];
  __amdgpu_buffer_rsrc_t rsrc;
  auto set_sources = [&]() {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int orow = 128 * (w & 1) + 8 * j + r8;  // waves 0, 1 stage the activation rows, 2, 3 the weight rows
      const int lc = p ^ ((orow >> 1) & 7);
      voff[j] = w < 2 ? (min(m0 + orow, a.M - 1) - m0) * a.ldx * 2 + lc * 16 : (min(n0 + orow, a.N - 1) - n0) * a.ldw * 2 + lc * 16;
    }
    const char* origin = w < 2 ? (const char*)a.X + (size_t)m0 * a.ldx * 2 : (const char*)a.W + (size_t)n0 * a.ldw * 2;
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, 0x7fffffff, 0x00020000);
  };
  const int dst0 = (w < 2 ? 0 : TILE) + (128 * (w & 1)) * 128;
  auto dma = [&](int t, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(smem + (t & 1) * STAGE + dst0 + j * 1024), 16, voff[j], t * 128, 0, 0);
  };
  auto issue_prologue = [&]() {
#pragma unroll
    for (int j = 0; j < 16; ++j) dma(0, j);
#pragma unroll
    for (int j = 0; j < 16; ++j) dma(1, j);
  };
  const int sw = (c >> 1) & 7;
  const int xrow0 = (128 * wr + c) * 128;
  const int wrow0 = TILE + (128 * wc + c) * 128;
  i32x4 xf[2][8], wf[2][8];
  auto read_x = [&](const char* st, int kk, int i) { xf[kk][i] = *(const i32x4*)(st + xrow0 + (16 * i) * 128 + (((kk * 4 + q) ^ sw) << 4)); };
  auto read_w = [&](const char* st, int kk, int j) { wf[kk][j] = *(const i32x4*)(st + wrow0 + (16 * j) * 128 + (((kk * 4 + q) ^ sw) << 4)); };

  origin_of(tile);
  set_sources();
  issue_prologue();
  int after = 0;  // VMEM operations of this wave issued after the prologue DMAs (the epilogue's loads and stores), or -1: unknown
  while (true) {
    // K tile 0 of this output tile has landed when only K tile 1's 16 DMAs and whatever was issued after them are outstanding
    // (vmcnt retires in issue order); edge tiles (stores predicated) wait for everything but the newest 16
    if (after == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (after == 32) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (after == 56) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SLOT();
#pragma unroll
    for (int i = 0; i < 8; ++i) { read_x(smem, 0, i); read_w(smem, 0, i); }
    SLOT();

    // one K tile: first (k-step 0) and second (k-step 1) half of 64 MFMAs each; DMA: tile t + 2 is requested (not for the last two);
    // FIRST: the accumulators' first touch
    auto ktile = [&](int t, auto dma_c, auto first_c) {
      constexpr bool DMA = decltype(dma_c)::value, FIRST = decltype(first_c)::value;
      const char* st = smem + (t & 1) * STAGE;
      const char* sn = smem + ((t + 1) & 1) * STAGE;
      for_each_slot(std::make_integer_sequence<int, 64>{}, [&](auto ic) {
        constexpr int s = decltype(ic)::value, j = s >> 3, i = s & 7;
        if constexpr (FIRST) acc_mfma_first(s, __builtin_bit_cast(bf16x8, wf[0][j]), __builtin_bit_cast(bf16x8, xf[0][i]));
        else acc_mfma(s, __builtin_bit_cast(bf16x8, wf[0][j]), __builtin_bit_cast(bf16x8, xf[0][i]));
        SLOT();
        if (s < 16) {
          if (s < 8) read_x(st, 1, s);
          else read_w(st, 1, s - 8);
          SLOT();
        }
        if (s == 24) {
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), visible to the compiler's wait bookkeeping
          __builtin_amdgcn_s_barrier();
          SLOT();
        }
        if (DMA && s >= 26 && ((s - 26) % 6) == 0) {
          dma(t + 2, (s - 26) / 6);  // DMAs 0..5 of tile t + 2 behind slots 26, 32, ..., 56 (one per six MFMAs: bunched - every third
          SLOT();                    // or fourth slot - they cost more issue time than the extra slack to land buys, measured)
        }
      });
      for_each_slot(std::make_integer_sequence<int, 64>{}, [&](auto ic) {
        constexpr int s = decltype(ic)::value, j = s >> 3, i = s & 7;
        acc_mfma(s, __builtin_bit_cast(bf16x8, wf[1][j]), __builtin_bit_cast(bf16x8, xf[1][i]));
        SLOT();
        if (DMA && (s % 6) == 2 && s <= 56) {
          dma(t + 2, 6 + s / 6);  // DMAs 6..15 behind slots 2, 8, ..., 56
          SLOT();
        }
        if (s == 12) {
          if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile t + 1 landed: 6 + 2 of tile t + 2 outstanding
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          SLOT();
        }
        if (s >= 13 && s < 29) {  // (behind the last K tile these read a stage nobody needs: no branch in the stream)
          constexpr int f = s - 13;
          if (f < 8) read_x(sn, 0, f);
          else read_w(sn, 0, f - 8);
          SLOT();
        }
      });
    };
    if (nk > 2) {
      ktile(0, std::true_type{}, std::true_type{});
      for (int t = 1; t < nk - 2; ++t) ktile(t, std::true_type{}, std::false_type{});
      ktile(nk - 2, std::false_type{}, std::false_type{});
    } else {
      ktile(0, std::false_type{}, std::true_type{});
    }
    ktile(nk - 1, std::false_type{}, std::false_type{});

    // ---- what the epilogue needs from global memory is requested before the next tile's DMAs (gemm256.hip)
    const int em0 = m0, en0 = n0;
    bf16x4 bv[8];
    if constexpr (!is_glu<EPI>) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int n = en0 + 128 * wc + 16 * nt + 4 * q;
        bv[nt] = (a.bias && n < a.N) ? *(const bf16x4*)(a.bias + n) : bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      }
    }
    // residual rows in the order the 8 store rounds consume them (round = 4 ch + pass: 32 rows x 64 columns), two rounds ahead
    bf16x8 rs[2][4];
    auto load_res = [&](int round) {
      const int ch = round >> 2, pass = round & 3;
      const int n = en0 + 128 * wc + 64 * ch + 8 * (lane & 7);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = em0 + 128 * wr + 32 * pass + 8 * i + (lane >> 3);
        rs[round & 1][i] = *(const bf16x8*)(a.res + (size_t)min(m, a.M - 1) * a.ldres + min(n, a.N - 8));
      }
    };
    if constexpr (EPI == EPI_RESIDUAL) {
      load_res(0);
      load_res(1);
    }
    const bool interior = em0 + BM <= a.M && en0 + BN <= a.N;
    tile += gridDim.x;
    const bool more = tile < ntiles;  // workgroup-uniform
    if (more) {
      origin_of(tile);
      set_sources();
      issue_prologue();
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // the last MFMAs have left the pipe before the accumulators are read
    if constexpr (is_glu<EPI>) {
      for_each_slot(std::make_integer_sequence<int, 32>{}, [&](auto ic) {
        constexpr int k = decltype(ic)::value, mt = k >> 2, np = k & 3;  // gate tile 2 np, up tile 2 np + 1
        store_glu<EPI>(a, acc_read(8 * (2 * np) + mt), acc_read(8 * (2 * np + 1) + mt), em0 + 128 * wr + 16 * mt + c, en0 + 128 * wc + 32 * np, q);
      });
    } else {
      // Eight rounds of 32 rows x 64 columns through a staging region PRIVATE to the wave, double-buffered (2 x 4 KiB): round r + 1 is
      // converted and written while round r's rows are on their way back from LDS - a lone wave per SIMD has no partner to cover
      // its LDS round trips (two dependent ones per round in the single-buffer form: 13 us of a 41-us fc1 tile sat outside the K loop)
      char* ep = smem + 2 * STAGE + w * 8192;
      const int pch = lane & 7;
      auto stage_round = [&](auto rc) {
        constexpr int round = decltype(rc)::value, ch = round >> 2, pass = round & 3;
        char* eb = ep + (round & 1) * 4096;
        for_each_slot(std::make_integer_sequence<int, 8>{}, [&](auto tc) {
          constexpr int k = decltype(tc)::value, mh = k >> 2, nt = k & 3, mt = 2 * pass + mh;
          const int ml = 16 * mh + c;
          const bf16x4 o = epi_act4<EPI>(acc_read(8 * (4 * ch + nt) + mt), bv[4 * ch + nt]);
          *(bf16x4*)(eb + ml * 128 + (((2 * nt + (q >> 1)) ^ (ml & 7)) << 4) + (q & 1) * 8) = o;
        });
      };
      stage_round(std::integral_constant<int, 0>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      for_each_slot(std::make_integer_sequence<int, 8>{}, [&](auto rc) {
        constexpr int round = decltype(rc)::value, ch = round >> 2, pass = round & 3;
        const int n = en0 + 128 * wc + 64 * ch + 8 * pch;
        const char* eb = ep + (round & 1) * 4096;
        bf16x8 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + (lane >> 3);
          v[i] = *(const bf16x8*)(eb + row * 128 + ((pch ^ (row & 7)) << 4));
        }
        if constexpr (round < 7) stage_round(std::integral_constant<int, round + 1>{});  // into the other buffer, under the reads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this round's rows are back, the next round's are written
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 32 * pass + 8 * i + (lane >> 3);
          const int m = em0 + 128 * wr + row;
          if (m < a.M && n < a.N) {
            if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[i][e] = f2bf(bf2f(v[i][e]) + bf2f(rs[round & 1][i][e]));
            }
            *(bf16x8*)(a.out + (size_t)m * a.ldo + n) = v[i];
          }
        }
        if constexpr (EPI == EPI_RESIDUAL)
          if (round < 6) load_res(round + 2);
      });
    }
    if (!more) break;
    after = !interior ? -1 : (EPI == EPI_RESIDUAL ? 56 : 32);
  }
}

// one workgroup per CU the calling thread may count on (hwocr_set_cu_budget), each walking tiles b, b + grid, ...
inline int w4_grid_cap() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int budget = hwocr_cu_budget();
  return budget > 0 && budget < cus ? budget : cus;
}
template <int EPI>
void launch_w4(const WideArgs& b, hipStream_t st) {
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)gemm_wide256w4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    return true;
  }();
  (void)done;
  const int ntiles = b.tilesM * b.tilesN, n = w4_grid_cap();
  hipLaunchKernelGGL((gemm_wide256w4_kernel<EPI>), dim3(ntiles < n ? ntiles : n), dim3(256), W4_LDS, st, b);
}

#ifdef HWOCR_DIAG
#include "diag_src/gemm256w4_experiments.inc"  // measurement variants + cycle stamps (tools/bench_gemm_w4.py); never in the product library
#endif

}  // namespace

// the four-wave form of hwocr_gemm_wide256 (called from gemm256.hip's launcher): bf16, every epilogue but the fused vision QKV; b has
// tilesM / tilesN filled in.  Returns false when the shape does not qualify (the caller launches the eight-wave kernel).
bool hwocr_gemm_wide256_w4(const gemm::WideArgs& b, int epi, bool forced, hipStream_t st) {
  if ((b.K >> 6) < 2 || (b.K & 63)) return false;
  // Where it is the faster of the two (same-run A/B on the page-read shapes, profiles/r03z_gemm_w4_product.txt): the gated and the plain
  // epilogues at any K (+3..5 %), bias + residual / activation ones up to K = 2048 (+0..2.5 %) and behind a longer K loop when the
  // output is at least 1536 wide (the decoders' down projections: +2..4.5 %).  A long K loop over FEW column tiles streams a big
  // activation panel with little reuse (the tower's fc2, 62208 x 1280 x 5120: 5 column tiles): there this kernel's shallower prefetch -
  // tile t + 2 can only be requested once tile t has left its stage - loses 3 %, so those stay with the eight-wave kernel.
  // HWOCR_GEMM256=4 forces it everywhere, =2 nowhere.
  if (!forced && !(epi == EPI_LINEAR || epi == EPI_SWIGLU || epi == EPI_GEGLU || b.K <= 2048 || b.N >= 1536)) return false;
  if (hwocr_plan_on()) {
    const int tiles = b.tilesM * b.tilesN, grid = tiles < w4_grid_cap() ? tiles : w4_grid_cap();
    hwocr_plan_note("gemm_wide256w4_kernel<epi=%d> M=%d N=%d K=%d tiles=%d grid=%d rounds=%d ktiles=%d", epi, b.M, b.N, b.K, tiles, grid,
                    (tiles + grid - 1) / grid, b.K >> 6);
    return true;
  }
  switch (epi) {
    case EPI_LINEAR: launch_w4<EPI_LINEAR>(b, st); return true;
    case EPI_RESIDUAL: launch_w4<EPI_RESIDUAL>(b, st); return true;
    case EPI_QUICKGELU: launch_w4<EPI_QUICKGELU>(b, st); return true;
    case EPI_GELU: launch_w4<EPI_GELU>(b, st); return true;
    case EPI_GELU_TANH: launch_w4<EPI_GELU_TANH>(b, st); return true;
    case EPI_SWIGLU: launch_w4<EPI_SWIGLU>(b, st); return true;
    case EPI_GEGLU: launch_w4<EPI_GEGLU>(b, st); return true;
  }
  return false;
}

#ifdef HWOCR_DIAG

// out[M][N] = bf16(X[M][K] . W[N][K]^T + bias); K % 64 == 0, N % 8 == 0 (diagnostic entry point, not in hwocr.h)
extern "C" int hwocr_debug_gemm_w4(const void* X, const void* W, const void* bias, void* out, int M, int N, int K, hipStream_t stream) {
  static const int abl = [] { const char* e = getenv("HWOCR_W4_ABLATE"); return e ? atoi(e) : 0; }();
  (void)hipGetLastError();
  if (!X || !W || !out || M <= 0 || N <= 0 || K < 128 || (K % 64) || (N % 8)) return HWOCR_EINVAL;
  WideArgs a{(const bf16*)X, (const bf16*)W, (const bf16*)bias, nullptr, (bf16*)out, M, N, K, K, K, N, 0, (M + BM - 1) / BM, (N + BN - 1) / BN};
  static const int form = [] { const char* e = getenv("HWOCR_W4_FORM"); return e ? atoi(e) : 16; }();
  auto k32 = abl == 1 ? gemm_w4x32_kernel<1> : abl == 2 ? gemm_w4x32_kernel<2> : abl == 3 ? gemm_w4x32_kernel<3> : gemm_w4x32_kernel<0>;
  auto k = form == 32 ? k32 : abl == 1 ? gemm_w4_kernel<1> : abl == 2 ? gemm_w4_kernel<2> : abl == 3 ? gemm_w4_kernel<3> : abl == 4 ? gemm_w4_kernel<4>
           : abl == 6 ? gemm_w4_kernel<6> : gemm_w4_kernel<0>;
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)done;
  hipLaunchKernelGGL(k, dim3(a.tilesM * a.tilesN), dim3(256), LDS_BYTES, stream, a);
  return hwocr_launch_status();
}
// shader cycles / 100 MHz ticks of the main loop, per workgroup, of the last hwocr_debug_gemm_w4 launch
extern "C" int hwocr_debug_gemm_w4_stamps(unsigned long long* host, int n) {
  if (!host || n <= 0 || n > 2 * 8192) return HWOCR_EINVAL;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4_stamps), sizeof(unsigned long long) * n) == hipSuccess ? HWOCR_OK : HWOCR_ELAUNCH;
}
#endif  // HWOCR_DIAG
