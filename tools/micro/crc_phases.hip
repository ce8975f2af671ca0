// crc_phases.hip -- times the streaming checksum kernel variant by variant (checksum_kernel.h: bit mask V) and
// prints where a launch spends its time (in-kernel wall_clock64 stamps of workgroup-thread 0).
// Standalone: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../zlib-ng_amd/csrc crc_phases.hip -o bin/crc_phases
//   ./bin/crc_phases [MiB ...]        default sizes 64 256 1024
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define ZR_CRC_VARIANTS 1      // every tuning variant of the kernel, including the two timing-only ones
#include "checksum_kernel.h"
#include "tables.h"

using namespace zr;

namespace zr {
void set_error(const char *, ...) {}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(uint32_t *p, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31;
        p[i] = (uint32_t)(z * 0x94D049BB133111EBull >> 16);
    }
}

__global__ __launch_bounds__(1024) void null_kernel(Partial *p) {
    if (threadIdx.x == 0 && blockIdx.x == 1u << 30) p[0].pad = 1;
}

struct Run {
    const char *name;
    void (*launch)(StreamArgs, const DeviceTables *, Partial *, int groups, hipStream_t, hipEvent_t, hipEvent_t);
    bool crc, adler, profile;
};

// e0 / e1 non-null: events attached to the dispatch itself (hipExtLaunchKernelGGL), i.e. the kernel's own timestamps
template <bool A, bool C, int V, bool P, int U = 4>
static void launcher(StreamArgs sa, const DeviceTables *t, Partial *p, int groups, hipStream_t st, hipEvent_t e0,
                     hipEvent_t e1) {
    hipExtLaunchKernelGGL((stream_kernel<A, C, false, V, P, U>), dim3(groups), dim3(kWgThreads), 0, st, e0, e1, 0, sa, t, p);
}

int main(int argc, char **argv) {
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back((size_t)atol(argv[i]) << 20);
    if (sizes.empty()) sizes = {64u << 20, 256u << 20, 1024u << 20};
    const size_t maxn = *std::max_element(sizes.begin(), sizes.end());
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.gcnArchName, cus);

    uint8_t *buf;
    CK(hipMalloc(&buf, maxn + 4096));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, (maxn + 4096) / 4);
    static DeviceTables host_tabs;
    build_tables(host_tabs);
    DeviceTables *tabs;
    CK(hipMalloc(&tabs, sizeof(DeviceTables)));
    CK(hipMemcpy(tabs, &host_tabs, sizeof(DeviceTables), hipMemcpyHostToDevice));
    Partial *partials;
    CK(hipMalloc(&partials, sizeof(Partial) * kMaxGroups));
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, 8 * sizeof(unsigned long long) * kMaxGroups));
    uint32_t *out;
    CK(hipMalloc(&out, 16));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    const Run runs[] = {
        {"adler32 (product)", launcher<true, false, kCrcVariant, false, kRowsPerBuffer>, false, true, false},
        {"adler32, four rows per buffer", launcher<true, false, kCrcVariant, false, 4>, false, true, false},
        {"crc32 V=0, four rows (round-1 form)", launcher<false, true, 0, false, 4>, true, false, false},
        {"crc32 V=194 built tables, lds barrier, x32 fold", launcher<false, true, 194, false, 4>, true, false, false},
        {"crc32 V=1218 + byte-addressed image", launcher<false, true, 1218, false, 4>, true, false, false},
        {"crc32 V=3266 + direct image", launcher<false, true, 3266, false, 4>, true, false, false},
        {"crc32 V=7362 + two-level epilogue", launcher<false, true, 7362, false, 4>, true, false, false},
        {"crc32 (product: V=7362, one row per buffer)", launcher<false, true, kCrcVariant, false, kRowsPerBuffer>, true, false, false},
        {"fused V=0, four rows (round-1 form)", launcher<true, true, 0, false, 4>, true, true, false},
        {"fused V=7362, four rows", launcher<true, true, 7362, false, 4>, true, true, false},
        {"fused (product)", launcher<true, true, kCrcVariant, false, kRowsPerBuffer>, true, true, false},
        {"crc32 (product) PROFILE", launcher<false, true, kCrcVariant, true, kRowsPerBuffer>, true, false, true},
        {"fused (product) PROFILE", launcher<true, true, kCrcVariant, true, kRowsPerBuffer>, true, true, true},
    };

    const int WARM = 300, REPS = 200;
    std::vector<hipEvent_t> ea(REPS), eb(REPS);
    for (int i = 0; i < REPS; ++i) {
        CK(hipEventCreate(&ea[i]));
        CK(hipEventCreate(&eb[i]));
    }
    for (size_t n : sizes) {
        printf("\n== %zu MiB ==\n", n >> 20);
        StreamArgs sa;
        sa.a0 = buf;
        sa.dst0 = nullptr;
        sa.n = (long long)n;
        sa.body = (long long)n;
        sa.nunits = (sa.body + kUnitBytes - 1) / kUnitBytes;
        sa.head = 0;
        sa.tail = 0;
        sa.phase_stamps = stamps;
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 8; ++i) {
                sa.bits.stride[k][i] = host_tabs.stride_tab[k][1u << i];
                sa.bits.x32[k][i] = host_tabs.x32_tab[k][1u << i];
            }
        int groups = cus;
        if (sa.nunits < groups) groups = (int)sa.nunits;
        FinalArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.tail_base = buf + n;
        fa.n = sa.n;
        fa.nunits = sa.nunits;
        fa.groups = groups;
        fa.adler_seed = 1;
        fa.crc_seed = 0;
        fa.crc_len_pow = xpow_bytes(host_tabs.pow_tab, (uint64_t)n);

        // launch floor: an empty kernel of the same shape
        for (int i = 0; i < WARM; ++i) hipLaunchKernelGGL(null_kernel, dim3(groups), dim3(kWgThreads), 0, st, partials);
        for (int i = 0; i < REPS; ++i) {
            CK(hipEventRecord(ea[i], st));
            hipLaunchKernelGGL(null_kernel, dim3(groups), dim3(kWgThreads), 0, st, partials);
            CK(hipEventRecord(eb[i], st));
        }
        CK(hipStreamSynchronize(st));
        {
            double sum = 0;
            for (int i = 0; i < REPS; ++i) {
                float ms;
                CK(hipEventElapsedTime(&ms, ea[i], eb[i]));
                sum += ms;
            }
            printf("%-46s %8.2f us\n", "empty kernel, same grid (event floor)", sum / REPS * 1e3);
        }
        uint32_t ref_crc = 0, ref_adler = 0;
        for (const Run &r : runs) {
            fa.do_adler = r.adler;
            fa.do_crc = r.crc;
            auto step = [&]() {
                r.launch(sa, tabs, partials, groups, st, nullptr, nullptr);
                hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, fa, tabs, partials, out, out + 1);
            };
            for (int i = 0; i < WARM; ++i) step();
            // (1) an event pair recorded in front of and behind the launch (round 1's trace_mark)
            for (int i = 0; i < REPS; ++i) {
                CK(hipEventRecord(ea[i], st));
                r.launch(sa, tabs, partials, groups, st, nullptr, nullptr);
                CK(hipEventRecord(eb[i], st));
                hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, fa, tabs, partials, out, out + 1);
            }
            CK(hipStreamSynchronize(st));
            double sum_pair = 0;
            for (int i = 0; i < REPS; ++i) {
                float v;
                CK(hipEventElapsedTime(&v, ea[i], eb[i]));
                sum_pair += v;
            }
            // (2) the events attached to the dispatch: the kernel's own start / stop
            for (int i = 0; i < REPS; ++i) {
                r.launch(sa, tabs, partials, groups, st, ea[i], eb[i]);
                hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, fa, tabs, partials, out, out + 1);
            }
            CK(hipStreamSynchronize(st));
            std::vector<float> ms(REPS);
            double sum = 0;
            for (int i = 0; i < REPS; ++i) {
                CK(hipEventElapsedTime(&ms[i], ea[i], eb[i]));
                sum += ms[i];
            }
            // (3) throughput: REPS steps back to back between two events
            CK(hipEventRecord(ea[0], st));
            for (int i = 0; i < REPS; ++i) step();
            CK(hipEventRecord(eb[0], st));
            CK(hipStreamSynchronize(st));
            float loop_ms;
            CK(hipEventElapsedTime(&loop_ms, ea[0], eb[0]));
            std::sort(ms.begin(), ms.end());
            uint32_t h[2];
            CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
            const double us = sum / REPS * 1e3;
            const char *verdict = "";
            if (r.crc) {
                if (!ref_crc) ref_crc = h[1];
                verdict = h[1] == ref_crc ? " crc ok" : " crc DIFFERS (expected for timing-only variants)";
            }
            if (r.adler) {
                if (!ref_adler) ref_adler = h[0];
                if (h[0] != ref_adler) verdict = " ADLER DIFFERS";
            }
            printf("%-46s dispatch %7.2f us (median %7.2f) %5.3f of 8 TB/s | event pair %7.2f | step in a loop %7.2f%s\n",
                   r.name, us, ms[REPS / 2] * 1e3, (double)n / 1e9 / (us * 1e-6) / 8000.0, sum_pair / REPS * 1e3,
                   loop_ms / REPS * 1e3, verdict);
            if (r.profile) {
                std::vector<unsigned long long> hs(8 * (size_t)groups);
                CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t_end = 0;
                for (int g = 0; g < groups; ++g) {
                    t0 = std::min(t0, hs[8 * g]);
                    t_end = std::max(t_end, hs[8 * g + 5]);
                }
                // wall_clock64 ticks at 100 MHz: 10 ns per tick
                static const char *names[6] = {"start", "tables staged", "image built", "main loop done", "lane products done", "partial written"};
                printf("    kernel span first start -> last end: %.2f us; per workgroup, us after the first start (min / median / max):\n",
                       (double)(t_end - t0) / 100.0);
                for (int k = 0; k < 6; ++k) {
                    std::vector<double> v(groups);
                    for (int g = 0; g < groups; ++g) v[g] = (double)(hs[8 * g + k] - t0) / 100.0;
                    std::sort(v.begin(), v.end());
                    printf("      %-20s %6.2f / %6.2f / %6.2f\n", names[k], v[0], v[groups / 2], v[groups - 1]);
                }
            }
        }
    }
    return 0;
}
