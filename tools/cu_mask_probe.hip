// GPU-box probe: which compute units does a stream created with hipExtStreamCreateWithCUMask really get?  (tdnet_opts.cu_reserve puts the
// Winograd GEMMs and their transforms on two queues with disjoint CU sets; the mask is a bit vector whose bit -> CU mapping the HIP headers do
// not document.  The KFD deals bit i to XCD i mod 8, then round-robin over the XCD's shader engines.)  For each mask a kernel of 8192 short
// workgroups stamps (XCC_ID, SE_ID, SH_ID, CU_ID) from HW_ID; the host prints the distinct CUs per XCD.  Also: two kernels on two masked
// streams at once -- do they overlap in time (disjoint masks) and does the masked kernel take proportionally longer?
// Build: hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o tools/_build/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_stamp(unsigned* out, int spin) {
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20) + (v == 1.25f);   // HW_REG_XCC_ID
    }
}
// a compute-bound kernel of `grid` persistent workgroups: time on a masked stream vs unmasked
__global__ void __launch_bounds__(256) k_burn(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.25f; b = b * 0.99999f + 1e-6f; }
    if (a == 123.f) out[0] = a + b;
}

static int report(const char* tag, hipStream_t s, unsigned* d, int nwg) {
    std::vector<unsigned> h(2 * nwg);
    hipLaunchKernelGGL(k_stamp, dim3(nwg), dim3(256), 0, s, d, 2000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<int, std::set<int>> per_xcc;
    for (int i = 0; i < nwg; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert(se * 64 + sh * 16 + cu);
    }
    int total = 0;
    printf("%-34s", tag);
    for (auto& kv : per_xcc) { printf(" xcd%d:%2zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total %d CUs\n", total);
    if (per_xcc.count(0)) {
        printf("    xcd0 (se.cu):");
        for (int id : per_xcc[0]) printf(" %d.%d", id / 64, id % 16);
        printf("\n");
    }
    return 0;
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int words = (ncu + 31) / 32, nwg = 8192;
    printf("device: %d compute units, %d mask words\n", ncu, words);
    unsigned* d = nullptr;
    CK(hipMalloc((void**)&d, 2 * nwg * 4));
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    if (report("no mask", plain, d, nwg)) return 1;
    struct M { const char* tag; std::vector<uint32_t> w; };
    std::vector<M> masks;
    auto low = [&](int lo, int hi) { std::vector<uint32_t> w(words, 0u); for (int i = lo; i < hi && i < ncu; ++i) w[i / 32] |= 1u << (i % 32); return w; };
    masks.push_back({"bits [0,32)", low(0, 32)});
    masks.push_back({"bits [32,256)", low(32, ncu)});
    masks.push_back({"bits [0,16)", low(0, 16)});
    masks.push_back({"bits [0,8)", low(0, 8)});
    masks.push_back({"bits [0,64)", low(0, 64)});
    { std::vector<uint32_t> w(words, 0u); for (int x = 0; x < words; ++x) w[x] = 0xFu; masks.push_back({"bits [32x,32x+4) of every word", w}); }
    { std::vector<uint32_t> w(words, 0u); w[0] = 0xFFFFFFFFu; masks.push_back({"word 0 only", w}); }
    std::vector<hipStream_t> st;
    for (auto& m : masks) {
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, m.w.data()));
        st.push_back(s);
        if (report(m.tag, s, d, nwg)) return 1;
    }
    // timing: compute-bound kernel with 4 workgroups per CU of the WHOLE chip, alone on: no mask / 224 CUs / 32 CUs; then 224 + 32 together
    float* o = nullptr;
    CK(hipMalloc((void**)&o, 64));
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    auto timed = [&](hipStream_t s, int grid, int iters, hipEvent_t a, hipEvent_t b) {
        hipEventRecord(a, s);
        hipLaunchKernelGGL(k_burn, dim3(grid), dim3(256), 0, s, o, iters);
        hipEventRecord(b, s);
    };
    for (int rep = 0; rep < 2; ++rep) {
        float t_plain, t_g, t_t, t_g2, t_t2;
        timed(plain, 4 * ncu, 200000, e0, e1); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&t_plain, e0, e1));
        timed(st[1], 4 * ncu, 200000, e0, e1); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&t_g, e0, e1));
        timed(st[0], 4 * 32, 200000, e0, e1); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&t_t, e0, e1));
        timed(st[1], 4 * ncu, 200000, e0, e1); timed(st[0], 4 * 32, 200000, f0, f1);
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&t_g2, e0, e1)); CK(hipEventElapsedTime(&t_t2, f0, f1));
        printf("burn: %d wgs unmasked %.3f ms | same grid on [32,256) %.3f ms (x%.3f; 256/224 = 1.143) | %d wgs on [0,32) %.3f ms | together: %.3f ms and %.3f ms\n",
               4 * ncu, t_plain, t_g, t_g / t_plain, 4 * 32, t_t, t_g2, t_t2);
    }
    return 0;
}
