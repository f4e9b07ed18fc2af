// tests/emu/tdemu.cpp -- TEST-ONLY fiber runtime behind tests/emu/td_device.h (see that header).
#include "td_device.h"

#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// ---- "device" memory and events -----------------------------------------------------------------------------------
hipError_t hipMalloc(void** p, size_t bytes) {
    void* q = nullptr;
    if (posix_memalign(&q, 256, bytes ? bytes : 256)) return 1;
    memset(q, 0xCD, bytes);                      // poison: reading uninitialised "HBM" shows up as huge values
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new tdemu_event{0.0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

// ---- context switch (x86-64 SysV: callee-saved registers + stack pointer) -----------------------------------------
extern "C" void tdemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl tdemu_switch
.type tdemu_switch,@function
tdemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size tdemu_switch,.-tdemu_switch
)");

namespace tdemu {
thread_local Fiber* cur = nullptr;
thread_local Idx g_blockIdx, g_gridDim, g_blockDim;
thread_local char* g_lds = nullptr;

static constexpr size_t STACK_BYTES = 96 * 1024;
static constexpr size_t LDS_BYTES = 160 * 1024;
static constexpr int MAX_THREADS = 1024;                  // 768: eight matrix waves + four loader waves (k_conv_dma_h3p)

struct WaveCtx {
    float a[2][64], b[2][64];
    float a8[2][64][8], b8[2][64][8];           // fp16 MFMA operands (widened)
    unsigned count = 0, gen = 0;
};
struct Worker {
    std::unique_ptr<char[]> stacks, lds;          // uninitialised: pages are committed only when a fiber touches them
    Fiber fibers[MAX_THREADS];
    WaveCtx waves[MAX_THREADS / 64];
    void* sched_sp = nullptr;
    int nthreads = 0, live = 0;
    unsigned bar_count = 0, bar_gen = 0;
    const std::function<void()>* body = nullptr;
    Worker() : stacks(new char[STACK_BYTES * MAX_THREADS]), lds(new char[LDS_BYTES + 256]) {}
};
static thread_local Worker* W = nullptr;

// switch from the current fiber to the next live one (round robin)
static void yield() {
    Fiber* me = cur;
    int i = (int)(me - W->fibers);
    for (;;) {
        i = (i + 1 == W->nthreads) ? 0 : i + 1;
        if (!W->fibers[i].done) break;
    }
    Fiber* nx = &W->fibers[i];
    if (nx == me) return;
    cur = nx;
    tdemu_switch(&me->sp, nx->sp);
}
// the same inside the current fiber's wave: lanes waiting in a wave collective only need their 63 wave-mates to arrive, so there is no
// point in visiting the other waves' fibers (which sit in their own collectives) -- 4x fewer context switches per MFMA at 256 threads
static void yield_in_wave() {
    Fiber* me = cur;
    const int idx = (int)(me - W->fibers), base = idx & ~63;
    const int top = (base + 64 < W->nthreads) ? base + 64 : W->nthreads;
    int i = idx;
    for (;;) {
        i = (i + 1 == top) ? base : i + 1;
        if (i == idx || !W->fibers[i].done) break;
    }
    Fiber* nx = &W->fibers[i];
    if (nx == me) return;
    cur = nx;
    tdemu_switch(&me->sp, nx->sp);
}
static void fiber_entry() {
    (*W->body)();
    Fiber* me = cur;
    me->done = true;
    W->live--;
    if (W->live == 0) {
        void* dummy;
        tdemu_switch(&dummy, W->sched_sp);       // back to the scheduler for good
    }
    int i = (int)(me - W->fibers);
    for (;;) {
        i = (i + 1 == W->nthreads) ? 0 : i + 1;
        if (!W->fibers[i].done) break;
    }
    cur = &W->fibers[i];
    void* dummy;
    tdemu_switch(&dummy, cur->sp);
    __builtin_trap();
}
void syncthreads() {
    const unsigned g = W->bar_gen;
    if (++W->bar_count == (unsigned)W->nthreads) { W->bar_count = 0; W->bar_gen++; return; }
    while (W->bar_gen == g) yield();
}
static void wave_barrier(WaveCtx& w) {
    const unsigned g = w.gen;
    if (++w.count == 64) { w.count = 0; w.gen++; return; }
    while (w.gen == g) yield_in_wave();
}
f32x16 mfma32(float a, float b, f32x16 c) {
    Fiber* f = cur;
    WaveCtx& w = W->waves[f->tidx.x >> 6];
    const int lane = f->tidx.x & 63, slot = f->seq & 1;
    f->seq++;
    w.a[slot][lane] = a;
    w.b[slot][lane] = b;
    wave_barrier(w);
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        c[r] = fmaf(w.a[slot][row + 32], w.b[slot][col + 32], fmaf(w.a[slot][row], w.b[slot][col], c[r]));
    }
    return c;
}
// v_mfma_f32_32x32x16_f16: lane l holds 8 consecutive k (k-group l>>5) of row/column l&31; products are exact in fp32
f32x16 mfma32_f16(f16x8 a, f16x8 b, f32x16 c) {
    Fiber* f = cur;
    WaveCtx& w = W->waves[f->tidx.x >> 6];
    const int lane = f->tidx.x & 63, slot = f->seq & 1;
    f->seq++;
    for (int e = 0; e < 8; ++e) { w.a8[slot][lane][e] = (float)a[e]; w.b8[slot][lane][e] = (float)b[e]; }
    wave_barrier(w);
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float s = c[r];
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) s += w.a8[slot][row + 32 * h][e] * w.b8[slot][col + 32 * h][e];
        c[r] = s;
    }
    return c;
}
// v_mfma_f32_32x32x16_bf16: same maps as the fp16 form; an operand is 8 bf16 packed two per dword.  Products of two bf16 are exact in
// fp32; the 16 products of an output are summed in double here and rounded once with the accumulator (the hardware's internal order is
// not documented -- the tests that use this compare against fp64 with a tolerance, never bit for bit).
f32x16 mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) {
    Fiber* f = cur;
    WaveCtx& w = W->waves[f->tidx.x >> 6];
    const int lane = f->tidx.x & 63, slot = f->seq & 1;
    f->seq++;
    auto widen = [](unsigned bits16) { const unsigned u = bits16 << 16; float x; memcpy(&x, &u, 4); return x; };
    for (int e = 0; e < 8; ++e) {
        w.a8[slot][lane][e] = widen((a[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        w.b8[slot][lane][e] = widen((b[e >> 1] >> (16 * (e & 1))) & 0xffffu);
    }
    wave_barrier(w);
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        double s = 0.0;
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) s += (double)w.a8[slot][row + 32 * h][e] * (double)w.b8[slot][col + 32 * h][e];
        c[r] = (float)((double)c[r] + s);
    }
    return c;
}
float shfl_xor(float v, int mask) {
    Fiber* f = cur;
    WaveCtx& w = W->waves[f->tidx.x >> 6];
    const int lane = f->tidx.x & 63, slot = f->seq & 1;
    f->seq++;
    w.a[slot][lane] = v;
    wave_barrier(w);
    return w.a[slot][lane ^ mask];
}

static void run_block(Worker* wk, const std::function<void()>& body, int nthreads) {
    W = wk;
    wk->body = &body;
    wk->nthreads = nthreads;
    wk->live = nthreads;
    wk->bar_count = 0;
    for (auto& wv : wk->waves) wv.count = 0;
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = wk->fibers[t];
        f.tidx = {(unsigned)t, 0, 0};
        f.seq = 0;
        f.done = false;
        char* top = wk->stacks.get() + (size_t)(t + 1) * STACK_BYTES;
        top = (char*)((uintptr_t)top & ~(uintptr_t)15);
        void** sp = (void**)(top - 64);
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;            // r15 r14 r13 r12 rbx rbp
        sp[6] = (void*)&fiber_entry;                              // return address
        sp[7] = nullptr;
        f.sp = sp;
    }
    cur = &wk->fibers[0];
    tdemu_switch(&wk->sched_sp, cur->sp);
    cur = nullptr;
}

// TDEMU_PROF=<file>: one line per launch (geometry, wall seconds) appended to the file -- where does a test's time go?
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes) {
    static const char* prof = getenv("TDEMU_PROF");
    const auto t0 = std::chrono::steady_clock::now();
    struct Done { const char* f; std::chrono::steady_clock::time_point t0; dim3 g, b; size_t l; ~Done() { if (!f) return;
        if (FILE* o = fopen(f, "a")) { fprintf(o, "grid %ux%ux%u block %u lds %zu\t%.6f\n", g.x, g.y, g.z, b.x, l,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); fclose(o); } } } done{prof, t0, grid, block, lds_bytes};
    const int nthreads = (int)(block.x * block.y * block.z);
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nthreads > MAX_THREADS || lds_bytes > LDS_BYTES || block.y != 1 || block.z != 1) {
        fprintf(stderr, "tdemu: unsupported launch geometry\n");
        abort();
    }
    static std::vector<Worker*> pool;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int nw = (int)std::thread::hardware_concurrency();
    if (nw < 1) nw = 1;
    if (nw > 16) nw = 16;
    if (nw > nblocks) nw = (int)nblocks;
    while ((int)pool.size() < nw) pool.push_back(new Worker());
    std::vector<std::thread> th;
    for (int wi = 0; wi < nw; ++wi) {
        th.emplace_back([&, wi]() {
            Worker* wk = pool[wi];
            g_gridDim = {grid.x, grid.y, grid.z};
            g_blockDim = {block.x, 1, 1};
            g_lds = (char*)(((uintptr_t)wk->lds.get() + 255) & ~(uintptr_t)255);
            for (long b = wi; b < nblocks; b += nw) {
                g_blockIdx = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y))};
                memset(g_lds, 0xCD, lds_bytes);                  // LDS is not zero-initialised on hardware either
                run_block(wk, body, nthreads);
            }
        });
    }
    for (auto& t : th) t.join();
}
}  // namespace tdemu
