// td_device.h -- the (small) set of gfx950 device/runtime primitives the TDNet kernels are written against.
//
// Everything here is CDNA4-only: 64-lane wavefronts, the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32
// fma chain at the fp32 vector rate, 64 FLOP/clk/SIMD -- MI355X_MICROARCH.md "Matrix cores"), dynamic LDS.
// tests/emu/ carries a host-side stand-in for this one header so the SAME kernel sources can be executed lane by
// lane on a CPU to verify their index math before spending GPU time; the product library is built only from this file.
#ifndef TD_DEVICE_H
#define TD_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define TD_KERNEL __global__
#define TD_DEV __device__ __forceinline__
#define TD_HOSTDEV __host__ __device__ __forceinline__
#define TD_DEV_MEMBER static __device__ __forceinline__        // static member function of a device-side helper struct
#define TD_LAUNCH_BOUNDS(t, w) __launch_bounds__(t, w)
// all LDS is dynamic and 16-byte aligned (cdna_hip_programming.md Guideline 17)
#define TD_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
// every launch of the library goes through this macro and is counted (per host thread): a frame's launch count is a number on the bench line
static thread_local long td_launch_count = 0;
#define TD_LAUNCH(kern, grid, block, lds, stream, ...) do { ++td_launch_count; hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__); } while (0)

// D(32x32) += A(32x2) * B(2x32).  lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
TD_DEV f32x16 td_mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Raw buffer access (SRD built from kernel arguments = SGPRs).  A 16-byte load whose byte offset is >= the buffer size
// returns zeros: the hardware bounds check replaces the branch around zero-padded conv taps (offset TD_BUF_OOB).
struct TdBuf { __amdgpu_buffer_rsrc_t r; };
#define TD_BUF_OOB 0x80000000u
TD_DEV TdBuf td_make_buf(const float* p, unsigned bytes) {
    TdBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
    return b;
}
TD_DEV f32x4 td_buf_ld4(TdBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff_bytes, soff_bytes, 0));
}
TD_DEV f32x2 td_buf_ld2(TdBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(b.r, voff_bytes, soff_bytes, 0));
}
// 16-byte store; dropped by the hardware when voff_bytes is outside the buffer.  There is deliberately NO soffset parameter: a buffer
// store of more than 64 bits keeps reading its data VGPRs after issue, and a VALU write to them in the next cycles corrupts the
// stored value.  LLVM's hazard recognizer inserts the wait state only when the soffset field is NOT a register
// (GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the instruction is not using a register in the soffset field"),
// which does not hold on gfx950: with the wave-uniform address part in an SGPR soffset, k_wino4_out_c<4> stored wrong odd dwords in
// lanes 12-15 of every 16, non-deterministically (tools/wino_vw_probe.py, profiles/r03d_wino_vw4_fix_probe.txt; 4- and 8-byte stores
// and the emulator are right, a pause BEFORE the store does not help).  With the literal 0 the compiler sees the hazard and spaces the
// overwrite; a wave-uniform offset belongs in voff_bytes.
TD_DEV void td_buf_st4(TdBuf b, unsigned voff_bytes, f32x4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b.r, voff_bytes, 0, 0);
}
TD_DEV void td_buf_st2(TdBuf b, unsigned voff_bytes, unsigned soff_bytes, f32x2 v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), b.r, voff_bytes, soff_bytes, 0);
}
TD_DEV void td_buf_st1(TdBuf b, unsigned voff_bytes, unsigned soff_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, voff_bytes, soff_bytes, 0);
}
TD_DEV float td_buf_ld1(TdBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff_bytes, soff_bytes, 0));
}

// LDS-DMA: 16 bytes per lane from the buffer straight into LDS at lds_wave_base + 16 * lane (the destination is wave-uniform base +
// lane-linear offset by construction of the instruction; only the SOURCE offset is per lane).  An out-of-range source offset
// writes zeros.  Completion is tracked by vmcnt like any load; nothing orders it against ds_read but the issuing wave's own wait.
TD_DEV void td_buf_ld16_lds(TdBuf b, char* lds_wave_base, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes, soff_bytes, 0, 0);
}
// wait until at most n of this wave's vector-memory operations (LDS-DMA pieces included) are outstanding; n is a compile-time constant
#define TD_WAIT_VM_PIECES(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// the bare workgroup barrier: no vmcnt drain (a __syncthreads() would wait for every LDS-DMA in flight).  The wave's own LDS reads are
// waited for first (lgkmcnt(0)) and nothing is scheduled across it: a buffer read before the barrier may be overwritten by another
// wave's DMA right after it.
#define TD_BARRIER_RAW() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); \
                              __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)

// wave priority 0..3 for the SIMD's instruction arbiter (priority first, then age)
#define TD_SETPRIO(n) __builtin_amdgcn_s_setprio(n)

// s_sleep: park the wave for ~64*n cycles (n <= 127); used to de-phase co-resident workgroups
#define TD_SLEEP(n) __builtin_amdgcn_s_sleep(n)

// compile-time instruction interleave hint (LLVM SchedGroupMask: 0x8 MFMA, 0x100 DS read, 0x200 DS write, 0x20 VMEM read)
#define TD_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define TD_PIN(x) asm volatile("" : "+v"(x))               // a use of x right here: the wait for a pending load of x is placed at this point, once
#define TD_VGPR_FLOOR_STR(n) #n
#define TD_VGPR_FLOOR(n) asm volatile("" ::: "v" TD_VGPR_FLOOR_STR(n))   // the kernel claims VGPRs 0..n: an occupancy ceiling that also holds against OTHER kernels' waves
#define TD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)    // the instruction scheduler moves nothing across this point
#define TD_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // tell the compiler a wave-uniform value is one (SGPR, usable as soffset)

// fp16 inputs, fp32 accumulate: D(32x32) += A(32x16) * B(16x32); lane l supplies 8 consecutive k of row/column l&31 (k-group l>>5)
TD_DEV f32x16 td_mfma32_f16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// bf16 inputs, fp32 accumulate (v_mfma_f32_32x32x16_bf16, 16x the rate of td_mfma32): same operand / accumulator maps as td_mfma32_f16; an
// operand is 8 bf16 packed two per dword (element 2 i in the low half of dword i).  td_gemm_b3.h feeds it the three bf16 parts of fp32 values.
TD_DEV f32x16 td_mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) {
    typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
// two fp32 -> two bf16, round to nearest even (v_cvt_pk_bf16_f32); a in the low half
TD_DEV unsigned td_pk_bf16(float a, float b) {
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
// a - b as ONE single-lane-width v_sub_f32 the SLP vectoriser cannot pair into v_pk_add_f32 (a packed fp32 op beside MFMAs costs more than
// its two halves: MI355X_MICROARCH.md "price of one filler beside MFMAs")
TD_DEV float td_sub1(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

TD_DEV float td_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
// value of lane ^ 1 (DPP quad_perm [1,0,3,2]: VALU rate, no LDS crossbar)
TD_DEV float td_swap1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
// true on every lane if the predicate holds on any lane of the wave (result is wave-uniform: usable as a branch condition)
TD_DEV bool td_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }
// orders this wave's LDS writes before its later LDS reads by OTHER lanes of the same wave (no workgroup barrier needed: a wave's
// LDS instructions execute in program order; this only stops the compiler from moving them across)
TD_DEV void td_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// 2^x as the bare v_exp_f32 (1 ulp; results below 2^-126 flush to zero): exp2f() wraps it in a range fix-up for denormal results --
// a compare, a select, an add and a multiply per call -- which the softmax does not need (such terms are 1e-38 of a sum >= 1) and
// which sits in the MFMA stream of the attention kernels 16 times per key tile.
TD_DEV float td_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// ---- two runtime facts the host code asks the platform (tests/emu/td_device.h answers them for the emulator) ---------------------------------
// Do two HIP streams share ONE hardware queue?  Two 40-us spin kernels started together: ~45 us for the pair = two queues, 80+ us = one.  The better
// of two tries (a context switch on the host must not look like a shared queue).  Synchronises both streams with the host; `a` stays ordered behind
// everything this enqueued.  e0..e2: events of the caller (timing enabled).
__global__ void k_queue_probe_spin(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
static inline hipError_t td_streams_share_a_queue(hipStream_t a, hipStream_t x, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, bool* shared, float* pair_us) {
    float worst = 1e9f;
    hipError_t e;
    for (int rep = 0; rep < 2; ++rep) {
        if ((e = hipEventRecord(e0, a)) != hipSuccess || (e = hipStreamWaitEvent(x, e0, 0)) != hipSuccess) return e;
        TD_LAUNCH(k_queue_probe_spin, dim3(1), dim3(64), 0, a, 4000ull);
        TD_LAUNCH(k_queue_probe_spin, dim3(1), dim3(64), 0, x, 4000ull);
        if ((e = hipEventRecord(e1, a)) != hipSuccess || (e = hipEventRecord(e2, x)) != hipSuccess || (e = hipStreamWaitEvent(a, e2, 0)) != hipSuccess) return e;
        if ((e = hipEventSynchronize(e1)) != hipSuccess || (e = hipEventSynchronize(e2)) != hipSuccess) return e;
        float t1 = 0.f, t2 = 0.f;
        if ((e = hipEventElapsedTime(&t1, e0, e1)) != hipSuccess || (e = hipEventElapsedTime(&t2, e0, e2)) != hipSuccess) return e;
        const float m = t1 > t2 ? t1 : t2;
        worst = m < worst ? m : worst;
    }
    *shared = worst > 0.064f;                                          // 40 us each: 40-45 us side by side, 80+ us one after the other
    if (pair_us) *pair_us = worst * 1e3f;
    return hipSuccess;
}
// is this stream being captured into a hipGraph right now?
static inline bool td_stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
}

TD_DEV int td_lane() { return threadIdx.x & 63; }
TD_DEV int td_wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
#endif  // TD_DEVICE_H
