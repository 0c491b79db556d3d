// ta_microbench.hip -- what one vector-memory instruction costs the texture addresser / L1 on gfx950, by access shape.
// Development aid for lcp.hip (the verify kernel is bound by that unit at the metric size): every wave issues the same
// number of loads of one shape from a table that stays in L2 (or in the CU's L1), many waves per SIMD, and the kernel time
// gives cycles per wave-instruction per CU.   build: hipcc -O3 --offload-arch=gfx950 tools/ta_microbench.hip -o tools/bin/ta_microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// SHAPE: 0 coalesced 16 B/lane; 1 gather 16 B, 64 lines; 2 gather 8 B, 64 lines; 3 gather 4 B, 64 lines; 4 gather 16 B inside 8 lines (8 lanes per line);
//        5 gather 8 B inside 8 lines; 6 gather 8 B inside 4 lines; 7 gather 16 B, 64 lines, 19 of 64 lanes active; 8 gather 8 B, 16 lines (4 lanes per line);
//        9 gather 16 B, 64 lines, 10 of 64 lanes active; 10 gather 8 B, 2 lines
template <int SHAPE>
__device__ __forceinline__ float one_load(const char* __restrict__ tab, uint32_t lines_mask, uint32_t s, uint32_t lane, int i) {
    const uint32_t r = mix(s ^ (lane * 0x9E3779B9u));   // per-lane random
    const uint32_t g8 = mix(s ^ ((lane >> 3) * 0x85EBCA6Bu)), g16 = mix(s ^ ((lane >> 4) * 0xC2B2AE35u)), g4 = mix(s ^ ((lane >> 2) * 0x27D4EB2Fu)), g32 = mix(s ^ ((lane >> 5) * 0x165667B1u));
    if (SHAPE == 0) { const float4 v = *(const float4*)(tab + ((size_t)((s & lines_mask) & ~7u) * 128u) % ((size_t)(lines_mask + 1) * 128u - 1024u) + lane * 16); return v.x + v.w; }
    if (SHAPE == 1) { const float4 v = *(const float4*)(tab + (size_t)(r & lines_mask) * 128 + ((r >> 24) & 7) * 16); return v.x + v.w; }
    if (SHAPE == 2) { const float2 v = *(const float2*)(tab + (size_t)(r & lines_mask) * 128 + ((r >> 24) & 15) * 8); return v.x + v.y; }
    if (SHAPE == 3) { const float v = *(const float*)(tab + (size_t)(r & lines_mask) * 128 + ((r >> 24) & 31) * 4); return v; }
    if (SHAPE == 4) { const float4 v = *(const float4*)(tab + (size_t)(g8 & lines_mask) * 128 + (lane & 7) * 16); return v.x + v.w; }
    if (SHAPE == 5) { const float2 v = *(const float2*)(tab + (size_t)(g8 & lines_mask) * 128 + (lane & 7) * 8 + ((r >> 20) & 1) * 64); return v.x + v.y; }
    if (SHAPE == 6) { const float2 v = *(const float2*)(tab + (size_t)(g16 & lines_mask) * 128 + (lane & 15) * 8); return v.x + v.y; }
    if (SHAPE == 7) { if ((mix(lane + 77u * (uint32_t)i) % 64u) < 19u) { const float4 v = *(const float4*)(tab + (size_t)(r & lines_mask) * 128 + ((r >> 24) & 7) * 16); return v.x + v.w; } return 0.f; }
    if (SHAPE == 8) { const float2 v = *(const float2*)(tab + (size_t)(g4 & lines_mask) * 128 + ((r >> 24) & 15) * 8); return v.x + v.y; }
    if (SHAPE == 9) { if ((mix(lane + 77u * (uint32_t)i) % 64u) < 10u) { const float4 v = *(const float4*)(tab + (size_t)(r & lines_mask) * 128 + ((r >> 24) & 7) * 16); return v.x + v.w; } return 0.f; }
    if (SHAPE == 10) { const float2 v = *(const float2*)(tab + (size_t)(g32 & lines_mask) * 128 + ((r >> 24) & 15) * 8); return v.x + v.y; }
    return 0.f;
}

// eight independent loads per trip (a wave that waits for each load before it issues the next measures latency, not the unit)
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ tab, uint32_t lines_mask, int iters, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    float acc = 0.f;
    uint32_t s = mix(wave * 2654435761u + 12345u);
    for (int i = 0; i < iters; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = s * 1664525u + 1013904223u; v[u] = one_load<SHAPE>(tab, lines_mask, s, lane, i + u); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 12345.678f) out[wave] = acc;   // never true: keeps the loads
}

typedef void (*kern_t)(const char*, uint32_t, int, float*);

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const kern_t ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>};
    const char* names[] = {"coalesced 16B/lane (8 lines)", "gather 16B, 64 lines", "gather 8B, 64 lines", "gather 4B, 64 lines", "gather 16B in 8 lines", "gather 8B in 8 lines",
                           "gather 8B in 4 lines", "gather 16B, 64 lines, 19/64 lanes", "gather 8B in 16 lines", "gather 16B, 64 lines, 10/64 lanes", "gather 8B in 2 lines"};
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters\": %d, \"rows\": [\n", prop.name, cus, prop.clockRate / 1000, iters);
    bool first = true;
    for (int tsel = 0; tsel < 2; ++tsel) {
        const uint32_t lines = tsel == 0 ? (1u << 14) : (1u << 7);   // 2 MB (L2-resident, beyond L1) / 16 KB (L1-resident)
        char* tab; CHECK(hipMalloc(&tab, (size_t)lines * 128 + 4096)); CHECK(hipMemset(tab, 0, (size_t)lines * 128 + 4096));
        for (int sidx = 0; sidx < 11; ++sidx) {
            const int blocks = cus * 8;   // 32 waves per CU
            hipLaunchKernelGGL(ks[sidx], dim3(blocks), dim3(256), 0, 0, tab, lines - 1, 50, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(ks[sidx], dim3(blocks), dim3(256), 0, 0, tab, lines - 1, iters, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_cu = 32.0 * iters;
            printf("%s {\"table\": \"%s\", \"shape\": \"%s\", \"ms\": %.4f, \"ns_per_wave_instr_per_cu\": %.2f}", first ? " " : ",\n ", tsel == 0 ? "2MB" : "16KB", names[sidx], ms, ms * 1e6 / instr_per_cu);
            first = false;
        }
        CHECK(hipFree(tab));
    }
    printf("\n]}\n");
    return 0;
}
