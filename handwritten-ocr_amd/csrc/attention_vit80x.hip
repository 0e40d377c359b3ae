// The head_dim-80 vision-tower attention for long segments as ONE wave per SIMD (gfx950 / MI355X).  A translation unit of its
// own because it is built with -mllvm -amdgpu-mfma-vgpr-form (build.py): the score MFMAs must write arch VGPRs (the vector pipe
// reads them), while the output accumulators sit in AGPRs that only inline asm names.  Launched by attention.hip: launch_vit80.
#include "attention_args.h"
#include "common.h"
#include <cstdlib>
#include <type_traits>

using namespace hwocr_attn;

namespace {

// O^T accumulators live in AGPRs a0..a95 that only the asm statements below name (accumulator idx = 3 * query block + d-tile ->
// a[16 idx : 16 idx + 15]); every statement lists its 16 registers as clobbers so that the kernel descriptor counts them and the
// register allocator keeps clear of them.  The allocator itself uses no AGPRs here (arch VGPR demand < 256, MFMA builtins in
// VGPR form: build flag -amdgpu-mfma-vgpr-form), which tools/check_vit80x_asm.py verifies on the emitted code.
#define ACC_CLOB_0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define ACC_CLOB_1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define ACC_CLOB_2 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define ACC_CLOB_3 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define ACC_CLOB_4 "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79"
#define ACC_CLOB_5 "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
__device__ __forceinline__ void acc_zero(int idx, bf16x8 z) {
  switch (idx) {
    case 0: asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %0, 0" ::"v"(z) : ACC_CLOB_0); break;
    case 1: asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %0, 0" ::"v"(z) : ACC_CLOB_1); break;
    case 2: asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %0, 0" ::"v"(z) : ACC_CLOB_2); break;
    case 3: asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %0, 0" ::"v"(z) : ACC_CLOB_3); break;
    case 4: asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %0, 0" ::"v"(z) : ACC_CLOB_4); break;
    case 5: asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %0, 0" ::"v"(z) : ACC_CLOB_5); break;
  }
}
// O^T[idx] += V^T fragment x P fragment
__device__ __forceinline__ void acc_mfma(int idx, bf16x8 vfrag, bf16x8 pfrag) {
  switch (idx) {
    case 0: asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(vfrag), "v"(pfrag) : ACC_CLOB_0); break;
    case 1: asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(vfrag), "v"(pfrag) : ACC_CLOB_1); break;
    case 2: asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(vfrag), "v"(pfrag) : ACC_CLOB_2); break;
    case 3: asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" ::"v"(vfrag), "v"(pfrag) : ACC_CLOB_3); break;
    case 4: asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" ::"v"(vfrag), "v"(pfrag) : ACC_CLOB_4); break;
    case 5: asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" ::"v"(vfrag), "v"(pfrag) : ACC_CLOB_5); break;
  }
}
// O^T[idx] *= alpha (per lane).  The caller has padded with s_nop so that the last MFMA into the accumulator has retired.
__device__ __forceinline__ void acc_scale(int idx, float alpha) {
  float t0, t1;
  switch (idx) {
    case 0: asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a3\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %1\n\tv_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %1\n\tv_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a7\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %1\n\tv_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %1\n\tv_accvgpr_read_b32 %0, a10\n\tv_accvgpr_read_b32 %1, a11\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_write_b32 a11, %1\n\tv_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a12, %0\n\tv_accvgpr_write_b32 a13, %1\n\tv_accvgpr_read_b32 %0, a14\n\tv_accvgpr_read_b32 %1, a15\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_write_b32 a15, %1\n\ts_nop 3" : "=&v"(t0), "=&v"(t1) : "v"(alpha) : ACC_CLOB_0); break;
    case 1: asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a16, %0\n\tv_accvgpr_write_b32 a17, %1\n\tv_accvgpr_read_b32 %0, a18\n\tv_accvgpr_read_b32 %1, a19\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_write_b32 a19, %1\n\tv_accvgpr_read_b32 %0, a20\n\tv_accvgpr_read_b32 %1, a21\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a20, %0\n\tv_accvgpr_write_b32 a21, %1\n\tv_accvgpr_read_b32 %0, a22\n\tv_accvgpr_read_b32 %1, a23\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_write_b32 a23, %1\n\tv_accvgpr_read_b32 %0, a24\n\tv_accvgpr_read_b32 %1, a25\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a24, %0\n\tv_accvgpr_write_b32 a25, %1\n\tv_accvgpr_read_b32 %0, a26\n\tv_accvgpr_read_b32 %1, a27\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a26, %0\n\tv_accvgpr_write_b32 a27, %1\n\tv_accvgpr_read_b32 %0, a28\n\tv_accvgpr_read_b32 %1, a29\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a28, %0\n\tv_accvgpr_write_b32 a29, %1\n\tv_accvgpr_read_b32 %0, a30\n\tv_accvgpr_read_b32 %1, a31\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a30, %0\n\tv_accvgpr_write_b32 a31, %1\n\ts_nop 3" : "=&v"(t0), "=&v"(t1) : "v"(alpha) : ACC_CLOB_1); break;
    case 2: asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a32, %0\n\tv_accvgpr_write_b32 a33, %1\n\tv_accvgpr_read_b32 %0, a34\n\tv_accvgpr_read_b32 %1, a35\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_write_b32 a35, %1\n\tv_accvgpr_read_b32 %0, a36\n\tv_accvgpr_read_b32 %1, a37\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a36, %0\n\tv_accvgpr_write_b32 a37, %1\n\tv_accvgpr_read_b32 %0, a38\n\tv_accvgpr_read_b32 %1, a39\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_write_b32 a39, %1\n\tv_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a40, %0\n\tv_accvgpr_write_b32 a41, %1\n\tv_accvgpr_read_b32 %0, a42\n\tv_accvgpr_read_b32 %1, a43\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a42, %0\n\tv_accvgpr_write_b32 a43, %1\n\tv_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a44, %0\n\tv_accvgpr_write_b32 a45, %1\n\tv_accvgpr_read_b32 %0, a46\n\tv_accvgpr_read_b32 %1, a47\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a46, %0\n\tv_accvgpr_write_b32 a47, %1\n\ts_nop 3" : "=&v"(t0), "=&v"(t1) : "v"(alpha) : ACC_CLOB_2); break;
    case 3: asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a48, %0\n\tv_accvgpr_write_b32 a49, %1\n\tv_accvgpr_read_b32 %0, a50\n\tv_accvgpr_read_b32 %1, a51\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a50, %0\n\tv_accvgpr_write_b32 a51, %1\n\tv_accvgpr_read_b32 %0, a52\n\tv_accvgpr_read_b32 %1, a53\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a52, %0\n\tv_accvgpr_write_b32 a53, %1\n\tv_accvgpr_read_b32 %0, a54\n\tv_accvgpr_read_b32 %1, a55\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a54, %0\n\tv_accvgpr_write_b32 a55, %1\n\tv_accvgpr_read_b32 %0, a56\n\tv_accvgpr_read_b32 %1, a57\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a56, %0\n\tv_accvgpr_write_b32 a57, %1\n\tv_accvgpr_read_b32 %0, a58\n\tv_accvgpr_read_b32 %1, a59\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a58, %0\n\tv_accvgpr_write_b32 a59, %1\n\tv_accvgpr_read_b32 %0, a60\n\tv_accvgpr_read_b32 %1, a61\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a60, %0\n\tv_accvgpr_write_b32 a61, %1\n\tv_accvgpr_read_b32 %0, a62\n\tv_accvgpr_read_b32 %1, a63\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a62, %0\n\tv_accvgpr_write_b32 a63, %1\n\ts_nop 3" : "=&v"(t0), "=&v"(t1) : "v"(alpha) : ACC_CLOB_3); break;
    case 4: asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a64, %0\n\tv_accvgpr_write_b32 a65, %1\n\tv_accvgpr_read_b32 %0, a66\n\tv_accvgpr_read_b32 %1, a67\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_write_b32 a67, %1\n\tv_accvgpr_read_b32 %0, a68\n\tv_accvgpr_read_b32 %1, a69\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a68, %0\n\tv_accvgpr_write_b32 a69, %1\n\tv_accvgpr_read_b32 %0, a70\n\tv_accvgpr_read_b32 %1, a71\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a70, %0\n\tv_accvgpr_write_b32 a71, %1\n\tv_accvgpr_read_b32 %0, a72\n\tv_accvgpr_read_b32 %1, a73\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a72, %0\n\tv_accvgpr_write_b32 a73, %1\n\tv_accvgpr_read_b32 %0, a74\n\tv_accvgpr_read_b32 %1, a75\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a74, %0\n\tv_accvgpr_write_b32 a75, %1\n\tv_accvgpr_read_b32 %0, a76\n\tv_accvgpr_read_b32 %1, a77\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a76, %0\n\tv_accvgpr_write_b32 a77, %1\n\tv_accvgpr_read_b32 %0, a78\n\tv_accvgpr_read_b32 %1, a79\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a78, %0\n\tv_accvgpr_write_b32 a79, %1\n\ts_nop 3" : "=&v"(t0), "=&v"(t1) : "v"(alpha) : ACC_CLOB_4); break;
    case 5: asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a80, %0\n\tv_accvgpr_write_b32 a81, %1\n\tv_accvgpr_read_b32 %0, a82\n\tv_accvgpr_read_b32 %1, a83\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a82, %0\n\tv_accvgpr_write_b32 a83, %1\n\tv_accvgpr_read_b32 %0, a84\n\tv_accvgpr_read_b32 %1, a85\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a84, %0\n\tv_accvgpr_write_b32 a85, %1\n\tv_accvgpr_read_b32 %0, a86\n\tv_accvgpr_read_b32 %1, a87\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a86, %0\n\tv_accvgpr_write_b32 a87, %1\n\tv_accvgpr_read_b32 %0, a88\n\tv_accvgpr_read_b32 %1, a89\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a88, %0\n\tv_accvgpr_write_b32 a89, %1\n\tv_accvgpr_read_b32 %0, a90\n\tv_accvgpr_read_b32 %1, a91\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a90, %0\n\tv_accvgpr_write_b32 a91, %1\n\tv_accvgpr_read_b32 %0, a92\n\tv_accvgpr_read_b32 %1, a93\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a92, %0\n\tv_accvgpr_write_b32 a93, %1\n\tv_accvgpr_read_b32 %0, a94\n\tv_accvgpr_read_b32 %1, a95\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a94, %0\n\tv_accvgpr_write_b32 a95, %1\n\ts_nop 3" : "=&v"(t0), "=&v"(t1) : "v"(alpha) : ACC_CLOB_5); break;
  }
}
__device__ __forceinline__ f32x16 acc_read(int idx) {
  float f[16];
  switch (idx) {
    case 0: asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\tv_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\tv_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ACC_CLOB_0); break;
    case 1: asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\tv_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\tv_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\tv_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ACC_CLOB_1); break;
    case 2: asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\tv_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\tv_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\tv_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\tv_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\tv_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\tv_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ACC_CLOB_2); break;
    case 3: asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\tv_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\tv_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\tv_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\tv_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\tv_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ACC_CLOB_3); break;
    case 4: asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\tv_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\tv_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\tv_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\tv_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\tv_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ACC_CLOB_4); break;
    case 5: asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83\n\tv_accvgpr_read_b32 %4, a84\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a86\n\tv_accvgpr_read_b32 %7, a87\n\tv_accvgpr_read_b32 %8, a88\n\tv_accvgpr_read_b32 %9, a89\n\tv_accvgpr_read_b32 %10, a90\n\tv_accvgpr_read_b32 %11, a91\n\tv_accvgpr_read_b32 %12, a92\n\tv_accvgpr_read_b32 %13, a93\n\tv_accvgpr_read_b32 %14, a94\n\tv_accvgpr_read_b32 %15, a95" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]) : : ACC_CLOB_5); break;
  }
  f32x16 v;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = f[i];
  return v;
}

// ------------------------------------------------------------------------------------------------
// attn_vit80x: the same product for long segments as ONE wave per SIMD.  4 waves x 64 queries (two 32-query blocks per wave),
// the whole 512-register file per wave, and the overlap of the matrix and vector pipes arranged INSIDE the wave instead of
// left to three lockstep waves per SIMD (whose phases coincide: 17 % of SIMD cycles had both pipes busy):
//   phase A(t): S(t+1) = K(t+1).Q^T   (20 MFMAs)  beside  the weights exp2(S(t)c - m) of tile t still to do (20 of 32 pairs)
//   phase B(t): O += V^T(t).P(t)      (24 MFMAs)  beside  the row maximum of S(t+1), the new m / alpha, its first 12 pairs
// one MFMA then one vector group per gap, kept in place by sched_barrier(0); every K / V^T fragment feeds two MFMAs (both query
// blocks) and is read two fragments ahead.  K / V^T tiles: register-staged (buffer loads in the gaps of phase A, ds_write in those
// of phase B, two tiles ahead into a 4-deep ring of the LDS images of attn_vit80_kernel), one barrier per tile.  The running
// maximum is per query (a lane's m moves only when its tile maximum exceeds it by more than `slack`); O is rescaled after phase B
// when some lane's m moved.
// Measured (12 pages x 16 heads x 5184 tokens, N(0,1) data, tools/bench_vit80x_stamps.py): 2848 cycles per 64-key tile at 1.93 GHz
// = phase A 1196 + phase B 1295 + barrier 309 + glue 48; 2.07-2.12 ms per launch against 2.13-2.21 for the 12-wave form.
// (Folding the softmax scale into the queries and the running reference into the score MFMAs' initial accumulator - one v_exp per
// weight, no scale-and-subtract - ran 2553 cycles at 1.82 GHz, 2.02 ms, but the extra bf16 rounding of q cost up to 3.4 % of an
// output on rows with large scores (tests: test_attn_vit80_page_shape): dropped, the scores stay exact fp32 sums of bf16 products.)  A lone
// wave pays the SUM of its issue costs (MFMA 8, v_exp 8, other VALU 4, ds_read_b128 ~16, a 1-KiB buffer load ~60, an LDS-DMA
// piece ~100 even against an empty descriptor): 1408 cycles of matrix time per tile sit under ~2500 of issue.  Two waves per SIMD
// do not escape it: an 8-wave / 512-query form (query fragments re-read from LDS, <= 256 registers, waves 4..7 staggered by one
// phase) measured 5470 cycles per PAIR of tiles = the same per tile, 2.18-2.26 ms, and was dropped.
// ------------------------------------------------------------------------------------------------
constexpr int V80X_STAGES = 4;

__global__ __launch_bounds__(256, 1) void attn_vit80x_kernel(PrefillArgs a) {
  constexpr int HD = 80, NT = 256, NSTG = V80X_STAGES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int seg, h, q0;
  {
    const int npairs = a.heads * a.nseg, L = blockIdx.x;
    int pair, qb;
    if ((npairs & 7) == 0) {  // whole (head, page) pairs per XCD, as attn_vit80_kernel
      const int slot = L >> 3;
      pair = (slot / a.qblocks) * 8 + (L & 7);
      qb = slot % a.qblocks;
    } else {
      pair = L / a.qblocks;
      qb = L % a.qblocks;
    }
    seg = pair / a.heads;
    h = pair % a.heads;
    q0 = qb * 256;
  }
  const int len = a.lens[seg];
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const bf16* Qp = a.Q + seg * a.q_seg + h * a.q_head;
  const bf16* Kp = a.K + seg * a.k_seg + h * a.k_head;
  const bf16* Vp = a.VT + seg * a.v_seg + h * a.v_head;
  const int qi0 = q0 + 64 * w + r;                            // query of block 0; block 1 = + 32
  const int rk = (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1);  // tile row -> key permutation (bits 2,3 swapped)

  bf16x8 qf[2][5];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const bf16* qrow = Qp + (long)min(qi0 + 32 * qb, len - 1) * a.q_row + 8 * hh;
#pragma unroll
    for (int s = 0; s < 5; ++s) qf[qb][s] = *(const bf16x8*)(qrow + 16 * s);
  }
  // V^T rows 80..87 of every stage: row 80 = 1.0, the rest 0 (never touched by the DMA)
  for (int i = tid; i < NSTG * 8 * 64; i += NT) {
    const int stg = i >> 9, rr = (i >> 6) & 7, col = i & 63;
    ((bf16*)(smem + stg * V80_STAGE + V80_K0 + V80_K1 + (80 + rr) * 128))[col] = (bf16)(rr == 0 ? 1.0f : 0.0f);
  }
  const int nt = (len + 63) >> 6;
  // 20 DMA instructions of 1 KiB per tile, 5 per wave: pieces k = w + 4 e.  K d 0..63 (k < 8: 8 key rows each), K d 64..79 (k = 8, 9:
  // 32 key rows each), V^T (k >= 10: 8 d rows each); piece k lands at stage + 1024 k.  Buffer loads: the lane's offset inside tile 0
  // is fixed (one VGPR per piece), the tile advances through the scalar offset, and the descriptors end at the segment's last key
  // row / last padded V^T column, so K rows past the segment arrive as zeros (their scores are masked) and a tile past the last one
  // may be requested without harm.
  const int kstep = 64 * (int)a.k_row * 2;  // bytes per key tile
#ifdef VIT80X_NO_TRAFFIC  // timing-only diagnostic: empty descriptors drop every DMA load in the range check; stream, waits, barriers stay
#define VIT80X_RECORDS(n) 0
#else
#define VIT80X_RECORDS(n) (n)
#endif
  const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, (short)0, VIT80X_RECORDS(len * (int)a.k_row * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rV =
      __builtin_amdgcn_make_buffer_rsrc((void*)Vp, (short)0, VIT80X_RECORDS((79 * (int)a.v_row + 64 * ((len + 63) >> 6)) * 2), 0x00020000);
  int dvoff[5];
#pragma unroll
  for (int e = 0; e < 5; ++e) {
    const int k = w + 4 * e;
    if (k < 8) {
      const int row = 8 * k + (lane >> 3);
      dvoff[e] = row * (int)a.k_row * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    } else if (k < 10) {
      const int row = 32 * (k - 8) + (lane >> 1);
      dvoff[e] = row * (int)a.k_row * 2 + 128 + (((lane & 1) ^ ((row >> 3) & 1)) << 4);
    } else {
      const int d = 8 * (k - 10) + (lane >> 3);
      dvoff[e] = d * (int)a.v_row * 2 + (((lane & 7) ^ ((d >> 1) & 7)) << 4);
    }
  }
  const bool e2_is_k = w < 2;  // piece 2 of a wave: k = w + 8
  const __amdgpu_buffer_rsrc_t r2 = e2_is_k ? rK : rV;
  // Register staging instead of LDS-DMA: one wave per SIMD has no partner to cover an instruction that is slow to ISSUE, and a
  // `buffer_load ... lds` piece held its wave ~100 cycles (even against empty descriptors: tools/bench_vit80x_stamps.py) - 500 cycles
  // per tile with the matrix pipe idle.  A piece is one buffer_load_dwordx4 into 4 VGPRs early in phase A and one ds_write_b128 late
  // in phase B of the same step (hipcc counts both: its own vmcnt ladder); the LDS image is what the DMA wrote (lane-linear 1 KiB).
  bf16x8 stg[5];
  auto load_piece = [&](int e, int t) {
    const bool is_k = e < 2 || (e == 2 && e2_is_k);
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(e < 2 ? rK : (e == 2 ? r2 : rV), dvoff[e], is_k ? t * kstep : t * 128, 0);
    stg[e] = __builtin_bit_cast(bf16x8, v);
  };
  auto write_piece = [&](int e, int t) {
    *(bf16x8*)(smem + (t % NSTG) * V80_STAGE + (w + 4 * e) * 1024 + lane * 16) = stg[e];
  };
  auto stage_tile = [&](int t) {
#pragma unroll
    for (int e = 0; e < 5; ++e) load_piece(e, t);
#pragma unroll
    for (int e = 0; e < 5; ++e) write_piece(e, t);
  };

  // lane offsets of the operand fragments inside a stage (the bank swizzle depends on the row only through lane bits)
  int koff[5], voff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = rk * 128 + (((2 * s + hh) ^ ((rk >> 1) & 7)) << 4);  // key block 1: + 32 * 128
  koff[4] = V80_K0 + rk * 32 + ((hh ^ ((rk >> 3) & 1)) << 4);                                // key block 1: + 32 * 32
#pragma unroll
  for (int c = 0; c < 4; ++c) voff[c] = V80_K0 + V80_K1 + r * 128 + (((2 * c + hh) ^ ((r >> 1) & 7)) << 4);  // d-tile: + 4096

  {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    // an asm MFMA gets none of the compiler's hazard padding: z was written by the VALU instructions just before it (without the
    // pad the first accumulator came out non-finite on the MI355X: stale operand registers)
    asm volatile("s_nop 15" : "+v"(z));
#pragma unroll
    for (int idx = 0; idx < 6; ++idx) acc_zero(idx, z);
  }
  float m[2] = {NEG_BIG, NEG_BIG}, alpha[2] = {1.f, 1.f};
  const float c2 = a.scale_log2, slack = a.slack;

  f32x16 S0[2][2], S1[2][2];     // scores^T of the tile being weighted / the tile being scored: [query block][key block]
  bf16x8 P0[2][2][2], P1[2][2][2];  // weights as PV operands: [query block][key block][half]

  // one pair of weights: scores 2p, 2p+1 of group g = (query block, key block, half) -> one dword of the PV operand
  auto weigh = [&](auto& S, auto& P, int pair) {
    const int g = pair >> 2, p = pair & 3, qb = g >> 2, kb = (g >> 1) & 1, s2 = g & 1;
    const float x0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[qb][kb][8 * s2 + 2 * p], c2, -m[qb]));
    const float x1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[qb][kb][8 * s2 + 2 * p + 1], c2, -m[qb]));
    P[qb][kb][s2][2 * p] = f2bf(x0);
    P[qb][kb][s2][2 * p + 1] = f2bf(x1);
    // a finished fragment is pinned where it was computed: without a use in this block LLVM sinks the whole group of exponentials
    // past the next branch, next to the MFMA that reads it - out of the MFMA gaps they were placed in
    if (p == 3) asm volatile("" : "+v"(P[qb][kb][s2]));
  };
  // the row-maximum chain of one query block in 22 single-instruction steps
  float mx[2], mnew[2];
  auto maxstep = [&](auto& S, int qb, int k) {
    if (k == 0) mx[qb] = fmaxf(S[qb][0][0], S[qb][1][0]);
    else if (k < 16) mx[qb] = fmaxf(fmaxf(mx[qb], S[qb][0][k]), S[qb][1][k]);  // v_max3_f32
    else if (k == 16) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[qb]), __float_as_uint(mx[qb]), false, false);
      mx[qb] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * c2;  // both halves of a query column
    } else if (k == 17) {
      mnew[qb] = mx[qb] > m[qb] + slack ? mx[qb] : m[qb];
    } else if (k == 18) {
      alpha[qb] = __builtin_amdgcn_exp2f(m[qb] - mnew[qb]);
      m[qb] = mnew[qb];
    }
  };
  constexpr int MAXSTEPS = 19;
  auto mask_tail = [&](auto& S, int j0) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = j0 + 16 * (i >> 3) + 8 * hh + (i & 7);
        if (key >= len) S[qb][0][i] = -INFINITY;
        if (key + 32 >= len) S[qb][1][i] = -INFINITY;
      }
  };
  // phase A: scores of the tile in stage `st` into Sn, beside pairs 12..31 of (Sc -> Pc)
  auto phaseA = [&](const char* st, auto& Sn, auto& Sc, auto& Pc, bool weights, int tdma) {
    bf16x8 kf[3];
    auto kfrag = [&](int f) {  // f = 2 s + key block
      const int s = f >> 1, kb = f & 1;
      return *(const bf16x8*)(st + koff[s] + kb * (s < 4 ? 32 * 128 : 32 * 32));
    };
    kf[0] = kfrag(0);
    kf[1] = kfrag(1);
#pragma unroll
    for (int f = 0; f < 10; ++f) {
      if (f + 2 < 10) kf[(f + 2) % 3] = kfrag(f + 2);
      const int s = f >> 1, kb = f & 1;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if (s == 0) {
          f32x16 z;
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.f;
          Sn[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[f % 3], qf[qb][s], z, 0, 0, 0);
        } else {
          Sn[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[f % 3], qf[qb][s], Sn[qb][kb], 0, 0, 0);
        }
        // the MFMA leads its gap: left free, hipcc hoists a gap's vector work above its MFMA, two MFMAs end up adjacent and the
        // second one holds the (in-order) wave at issue for the 32 cycles of the first with no vector instruction going out
        __builtin_amdgcn_sched_barrier(0);
        if (weights) {
          weigh(Sc, Pc, 12 + 2 * f + qb);
          const int gap = 2 * f + qb;  // tile t + 2 is requested in gaps 2, 6, 10, 14, 18 ...
          if ((gap & 3) == 2) load_piece(gap >> 2, tdma);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // phase B: O += V^T.P of the tile in stage `st`, beside the maximum of Sn and pairs 0..11 of (Sn -> Pn)
  auto phaseB = [&](const char* st, auto& Pc, auto& Sn, auto& Pn, auto next_c, int tdma) {
    constexpr bool next = decltype(next_c)::value;
    bf16x8 vf[3];
    auto vfrag = [&](int f) {  // f = 3 (2 key block + half) + d-tile
      return *(const bf16x8*)(st + voff[f / 3] + (f % 3) * 4096);
    };
    vf[0] = vfrag(0);
    vf[1] = vfrag(1);
#pragma unroll
    for (int f = 0; f < 12; ++f) {
      if (f + 2 < 12) vf[(f + 2) % 3] = vfrag(f + 2);
      const int c = f / 3, d = f % 3;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        acc_mfma(3 * qb + d, vf[f % 3], Pc[qb][c >> 1][c & 1]);  // same accumulator again 6 gaps later
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (next) {
          const int gap = 2 * f + qb;
          if (gap < 10) {  // 38 chain steps over 10 gaps, the two query blocks alternating
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int id = 4 * gap + u;
              if (id < 2 * MAXSTEPS) maxstep(Sn, id & 1, id >> 1);
            }
          } else if (gap < 22) {
            weigh(Sn, Pn, gap - 10);
          }
          if (gap >= 13 && (gap & 1)) write_piece((gap - 13) >> 1, tdma);  // ... and written to its stage in gaps 13, 15 .. 21 of phase B
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto rescale = [&]() {
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // the last PV MFMAs have left the pipe before O is read
#pragma unroll
    for (int idx = 0; idx < 6; ++idx) acc_scale(idx, alpha[idx / 3]);
  };
  auto stage_ptr = [&](int t) -> char* { return smem + (t % NSTG) * V80_STAGE; };
  auto clear_tail_v = [&](int t) {
    // keys past the segment: their scores are masked, their V^T columns are whatever the buffer holds -> clear them in LDS
    char* st = stage_ptr(t);
    const int j0 = t * 64;
    for (int i = tid; i < 80 * 64; i += NT) {
      const int d = i >> 6, col = i & 63;
      if (j0 + col >= len) {
        const int ch = (col >> 3) ^ ((d >> 1) & 7);
        ((bf16*)(st + V80_K0 + V80_K1 + d * 128 + ch * 16))[col & 7] = (bf16)0.0f;
      }
    }
    __syncthreads();
  };
  // one tile: Sc / Pc = scores and weights of tile t (maximum taken, pairs 0..11 done), Sn / Pn = those of tile t + 1
#ifdef VIT80X_STAMPS
  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = tlast, w_begin = wall_clock64();
#define STAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[k] += now_ - tlast; tlast = now_; }
#else
#define STAMP(k)
#endif
  auto step = [&](int t, auto& Sc, auto& Pc, auto& Sn, auto& Pn, auto next_c) {
    constexpr bool next = decltype(next_c)::value;  // tile t + 1 exists (the last tile is peeled)
    STAMP(3)
    // tile t + 2 is requested and written during this step, piece by piece in the MFMA gaps, into the stage of tile t - 2 (every wave
    // is past the barrier that ended its last read)
    if constexpr (next) {
      STAMP(0)
      phaseA(stage_ptr(t + 1), Sn, Sc, Pc, true, t + 2);
      STAMP(1)
      if (t + 2 == nt && (len & 63)) {
        asm volatile("; tail tile");
        mask_tail(Sn, (t + 1) * 64);
      }
    } else {
#pragma unroll
      for (int pr = 12; pr < 32; ++pr) weigh(Sc, Pc, pr);
    }
    if (!next && (len & 63)) clear_tail_v(t);
    const float m0 = m[0], m1 = m[1];
    STAMP(0)
    phaseB(stage_ptr(t), Pc, Sn, Pn, next_c, t + 2);
    STAMP(2)
    if (next && __any(m[0] != m0 || m[1] != m1)) rescale();
    __syncthreads();  // tile t + 2 is in LDS (this step's ds_writes), every wave has finished with the stage of tile t
  };

  // ---- prologue: tiles 0..2 in flight, scores / maximum / first weights of tile 0 without overlap
  stage_tile(0);
  stage_tile(1);
  __syncthreads();
  phaseA(stage_ptr(0), S0, S1, P1, false, 0);
  if (nt == 1 && (len & 63)) mask_tail(S0, 0);
#pragma unroll
  for (int k = 0; k < MAXSTEPS; ++k) {
    maxstep(S0, 0, k);
    maxstep(S0, 1, k);
  }
#pragma unroll
  for (int pr = 0; pr < 12; ++pr) weigh(S0, P0, pr);
  // (O is still zero: the first alpha needs no rescale)
  int t = 0;
  for (; t + 2 < nt; t += 2) {
    step(t, S0, P0, S1, P1, std::true_type{});
    step(t + 1, S1, P1, S0, P0, std::true_type{});
  }
  if (nt - t == 2) {
    step(t, S0, P0, S1, P1, std::true_type{});
    step(t + 1, S1, P1, S0, P0, std::false_type{});
  } else {
    step(t, S0, P0, S1, P1, std::false_type{});
  }

#ifdef VIT80X_STAMPS
  STAMP(3)
  if (a.stamps && blockIdx.x < 64 && lane == 0) {  // cycles per wave: [0] glue, [1] phase A, [2] phase B, [3] rescale + wait + barrier + DMA issue
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) a.stamps[(blockIdx.x * 4 + w) * 6 + k2] = tacc[k2];
    a.stamps[(blockIdx.x * 4 + w) * 6 + 4] = __builtin_amdgcn_s_memtime() - t_begin;  // shader cycles and 100 MHz ticks of the tile loop
    a.stamps[(blockIdx.x * 4 + w) * 6 + 5] = wall_clock64() - w_begin;
  }
#endif
  // softmax denominator = accumulator row d = 80 (d-tile 2, local row 16 -> register 8 of the hh = 0 half)
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // (the loop ends on a barrier: the PV MFMAs are long done; kept for form)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    f32x16 o[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d] = acc_read(3 * qb + d);
    const float l = __shfl(o[2][8], r);
    const float inv = 1.0f / l;
    const int qi = qi0 + 32 * qb;
    if (qi < len) {
      bf16* orow = a.O + seg * a.o_seg + (long)qi * a.o_row + h * HD;
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = d * 32 + 8 * g + 4 * hh;
          if (dd < HD) {
            bf16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][4 * g + e] * inv);
            *(bf16x4*)(orow + dd) = ov;
          }
        }
    }
  }
}


}  // namespace

void hwocr_attn::launch_vit80x(const PrefillArgs& a, int grid, hipStream_t st) {
  static const bool done = [&] {  // thread-safe one-time setup: two lane threads reach a kernel's first launch together
    (void)hipFuncSetAttribute((const void*)attn_vit80x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, V80X_STAGES * V80_STAGE);
    return true;
  }();
  (void)done;
  hipLaunchKernelGGL(attn_vit80x_kernel, dim3(grid), dim3(256), V80X_STAGES * V80_STAGE, st, a);
}

#ifdef VIT80X_STAMPS
// Diagnostic build only (tools/bench_vit80x_stamps.py compiles this unit alone with -DVIT80X_STAMPS into a scratch library):
// the tower layout of the bench ([heads][rows][80] q / k, [heads][80][rows] v^T, pages of P rows), cycle totals per wave out.
extern "C" int vit80x_debug(const void* Q, const void* K, const void* VT, void* O, const int* lens, int nimg, int heads, int P, long rows,
                            float scale, float slack, unsigned long long* stamps, hipStream_t stream) {
  PrefillArgs a{(const bf16*)Q, (const bf16*)K, (const bf16*)VT, (bf16*)O, lens,
                (long)P * 80, rows * 80, 80, (long)P * 80, rows * 80, 80, (long)P, 80 * rows, rows, (long)P * heads * 80, (long)heads * 80,
                1, scale * 1.4426950408889634f, 0, heads, nimg, (P + 255) / 256};
  a.slack = slack;
  a.stamps = stamps;
  launch_vit80x(a, a.qblocks * heads * nimg, stream);
  return (int)hipGetLastError();
}
#endif
