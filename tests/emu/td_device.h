// tests/emu/td_device.h -- TEST-ONLY host stand-in for tdnet_amd/csrc/td_device.h.
//
// Purpose: execute the UNMODIFIED kernel sources (td_conv.h, td_attn.h, td_misc.h, td_model.hip) lane by lane on the
// CPU so their index math (LDS images, MFMA operand/accumulator maps, k-permutation, weight packing, epilogues, the
// whole per-frame orchestration) can be checked against the oracle in this GPU-less container before GPU minutes
// are spent.  It is compiled only into tests/emu/_build/libtdnet_emu.so by tests/emu/build_emu.py and loaded only by
// tests/test_emu_*.py.  Nothing under tdnet_amd/ references it; the product library is hipcc + the real header.
//
// Model: one workgroup = blockDim fibers (user-level contexts) run round-robin on one OS thread; __syncthreads and
// the wave-collective ops (MFMA, shuffle) are barriers over fibers.  The MFMA follows the documented gfx950 maps:
//   A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31], D reg r of lane l -> D[(r&3)+8(r>>2)+4(l>>5)][l&31],
//   result = fma(a_k1, b_k1, fma(a_k0, b_k0, c))   (cdna_hip_programming.md §3).
#ifndef TD_DEVICE_H   // same guard as the real header: the emu build force-includes this file first
#define TD_DEVICE_H
#define TD_EMU 1
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- HIP runtime stand-ins ("device" memory is host memory) ------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
typedef void* hipStream_t;
struct tdemu_event { double t; };
typedef tdemu_event* hipEvent_t;
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)0x1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)0x1; return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = (hipStream_t)0x1; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);

// the emulator has one "queue" and no graphs: streams never collide, nothing is ever being captured (csrc/td_device.h has the real answers)
inline hipError_t td_streams_share_a_queue(hipStream_t, hipStream_t, hipEvent_t, hipEvent_t, hipEvent_t, bool* shared, float* pair_us) {
    *shared = false;
    if (pair_us) *pair_us = 0.f;
    return hipSuccess;
}
inline bool td_stream_is_capturing(hipStream_t) { return false; }

// ---- execution model ------------------------------------------------------------------------------------------------
namespace tdemu {
struct Idx { unsigned x, y, z; };
struct Fiber {
    void* sp;
    Idx tidx;
    unsigned seq;       // wave-collective sequence number (double-buffer selector)
    bool done;
};
extern thread_local Fiber* cur;
extern thread_local Idx g_blockIdx, g_gridDim, g_blockDim;
extern thread_local char* g_lds;
void syncthreads();
f32x16 mfma32(float a, float b, f32x16 c);
f32x16 mfma32_f16(f16x8 a, f16x8 b, f32x16 c);
f32x16 mfma32_bf16(u32x4 a, u32x4 b, f32x16 c);
float shfl_xor(float v, int mask);
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes);
}  // namespace tdemu

#define threadIdx (tdemu::cur->tidx)
#define blockIdx (tdemu::g_blockIdx)
#define gridDim (tdemu::g_gridDim)
#define blockDim (tdemu::g_blockDim)
#define __syncthreads() tdemu::syncthreads()

#define TD_KERNEL static
#define TD_DEV static inline
#define TD_HOSTDEV static inline
#define TD_DEV_MEMBER static inline
#define TD_LAUNCH_BOUNDS(t, w)
#define TD_DYN_LDS(name) char* name = tdemu::g_lds
static thread_local long td_launch_count = 0;
#define TD_LAUNCH(kern, grid, block, lds, stream, ...) \
    do { ++td_launch_count; tdemu::launch([=]() { kern(__VA_ARGS__); }, grid, block, (size_t)(lds)); } while (0)

#define TD_SCHED_GROUP(mask, n) ((void)0)
#define TD_SCHED_FENCE() ((void)0)
#define TD_VGPR_FLOOR(n) ((void)0)
#define TD_PIN(x) ((void)0)
#define TD_UNIFORM(x) (x)
#define TD_SLEEP(n) ((void)0)
#define TD_SETPRIO(n) ((void)0)

struct TdBuf { const char* p; unsigned bytes; };
#define TD_BUF_OOB 0x80000000u
TD_DEV TdBuf td_make_buf(const float* p, unsigned bytes) { return TdBuf{(const char*)p, bytes}; }
TD_DEV f32x4 td_buf_ld4(TdBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned long long)voff_bytes + 16 <= b.bytes) {       // the hardware range check ignores soffset ...
        if ((unsigned long long)voff_bytes + soff_bytes + 16 > b.bytes) abort();   // ... so kernels must keep the sum in range themselves
        memcpy(&v, b.p + soff_bytes + voff_bytes, 16);
    }
    return v;
}
TD_DEV f32x2 td_buf_ld2(TdBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    f32x2 v = {0.f, 0.f};
    if ((unsigned long long)voff_bytes + 8 <= b.bytes) {       // the hardware range check ignores soffset ...
        if ((unsigned long long)voff_bytes + soff_bytes + 8 > b.bytes) abort();   // ... so kernels must keep the sum in range themselves
        memcpy(&v, b.p + soff_bytes + voff_bytes, 8);
    }
    return v;
}

TD_DEV void td_buf_st4(TdBuf b, unsigned voff_bytes, f32x4 v) {      // no soffset: see csrc/td_device.h
    if ((unsigned long long)voff_bytes + 16 <= b.bytes) memcpy(const_cast<char*>(b.p) + voff_bytes, &v, 16);
}
TD_DEV void td_buf_st2(TdBuf b, unsigned voff_bytes, unsigned soff_bytes, f32x2 v) {
    if ((unsigned long long)voff_bytes + 8 <= b.bytes) {
        if ((unsigned long long)voff_bytes + soff_bytes + 8 > b.bytes) abort();
        memcpy(const_cast<char*>(b.p) + soff_bytes + voff_bytes, &v, 8);
    }
}
TD_DEV void td_buf_st1(TdBuf b, unsigned voff_bytes, unsigned soff_bytes, float v) {
    if ((unsigned long long)voff_bytes + 4 <= b.bytes) {
        if ((unsigned long long)voff_bytes + soff_bytes + 4 > b.bytes) abort();
        memcpy(const_cast<char*>(b.p) + soff_bytes + voff_bytes, &v, 4);
    }
}
TD_DEV float td_buf_ld1(TdBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    float v = 0.f;
    if ((unsigned long long)voff_bytes + 4 <= b.bytes) {
        if ((unsigned long long)voff_bytes + soff_bytes + 4 > b.bytes) abort();
        memcpy(&v, b.p + soff_bytes + voff_bytes, 4);
    }
    return v;
}

// LDS-DMA stand-in: synchronous copy (zeros for an out-of-range source) to lds_wave_base + 16 * lane
TD_DEV void td_buf_ld16_lds(TdBuf b, char* lds_wave_base, unsigned voff_bytes, unsigned soff_bytes) {
    f32x4 v = td_buf_ld4(b, voff_bytes, soff_bytes);
    memcpy(lds_wave_base + 16 * (threadIdx.x & 63), &v, 16);
}
#define TD_WAIT_VM_PIECES(n) ((void)0)
#define TD_BARRIER_RAW() tdemu::syncthreads()

TD_DEV f32x16 td_mfma32(float a, float b, f32x16 c) { return tdemu::mfma32(a, b, c); }
TD_DEV f32x16 td_mfma32_f16(f16x8 a, f16x8 b, f32x16 c) { return tdemu::mfma32_f16(a, b, c); }
TD_DEV f32x16 td_mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) { return tdemu::mfma32_bf16(a, b, c); }
// fp32 -> bf16 bits, round to nearest even (NaN stays NaN): what v_cvt_pk_bf16_f32 does per element
TD_DEV unsigned tdemu_bf16_rne(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
TD_DEV float td_sub1(float a, float b) { return a - b; }
TD_DEV unsigned td_pk_bf16(float a, float b) { return tdemu_bf16_rne(a) | (tdemu_bf16_rne(b) << 16); }
TD_DEV float td_shfl_xor(float v, int mask) { return tdemu::shfl_xor(v, mask); }
TD_DEV float td_swap1(float v) { return tdemu::shfl_xor(v, 1); }
TD_DEV bool td_any(bool pred) {
    float f = pred ? 1.f : 0.f;
    for (int m = 1; m < 64; m <<= 1) { const float o = tdemu::shfl_xor(f, m); f = o > f ? o : f; }
    return f != 0.f;
}
TD_DEV void td_wave_sync() { (void)tdemu::shfl_xor(0.f, 1); }   // fibers of a wave meet here (a collective is the emulator's wave-level barrier)
TD_DEV float td_exp2(float x) { return exp2f(x); }
TD_DEV int td_lane() { return threadIdx.x & 63; }
TD_DEV int td_wave() { return threadIdx.x >> 6; }
#endif  // TD_DEVICE_H
