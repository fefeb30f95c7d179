// hbm_probe.hip -- measurement utility (not product code): attainable HBM READ bandwidth on gfx950
// for the access-pattern class of scv_hist_argmax (every workgroup streams whole contiguous 4 MiB
// cells with 16-byte loads), so roofline.frac can also be quoted against a measured read ceiling
// instead of the 8 TB/s datasheet number.  Usage: hbm_probe.bin [GB=40]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void read_cells(const v4i* __restrict__ src, long cell_vecs, long ncells, int* sink) {
    int acc = 0;
    for (long cell = blockIdx.x; cell < ncells; cell += gridDim.x) {
        const v4i* p = src + cell * cell_vecs;
        for (long i = threadIdx.x; i + (long)(U - 1) * blockDim.x < cell_vecs; i += (long)U * blockDim.x) {
            v4i x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(p + i + (long)u * blockDim.x) : p[i + (long)u * blockDim.x];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
        }
    }
    if (acc == 0x12345678) *sink = acc;
}

template <int U, bool NT>
__global__ void read_gridstride(const v4i* __restrict__ src, long nvec, int* sink) {
    int acc = 0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < nvec; i += U * stride) {
        v4i x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    if (acc == 0x12345678) *sink = acc;
}

__global__ void fill(v4i* dst, long nvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) { v4i v = {(int)i, 1, 2, 3}; dst[i] = v; }
}

template <typename F>
double time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 40.0;
    const long cell_bytes = 4l << 20;
    const long ncells = (long)(gb * 1e9 / cell_bytes);
    const long bytes = ncells * cell_bytes, nvec = bytes / 16, cell_vecs = cell_bytes / 16;
    v4i* buf; int* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    fill<<<4096, 256>>>(buf, nvec); CK(hipDeviceSynchronize());
    printf("buffer %.2f GB = %ld cells of 4 MiB\n", bytes / 1e9, ncells);
    struct Cfg { int grid, threads; };
    const Cfg cfgs[] = {{256, 1024}, {250, 1024}, {512, 512}, {500, 512}, {256, 512}, {1024, 256}, {2048, 256}};
    for (const Cfg& c : cfgs) {
        double a = time_ms([&] { read_cells<4, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double b = time_ms([&] { read_cells<4, false><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double d = time_ms([&] { read_cells<8, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double e = time_ms([&] { read_cells<2, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        printf("cells   grid %4d x %4d : U4 nt %6.0f | U4 plain %6.0f | U8 nt %6.0f | U2 nt %6.0f GB/s\n", c.grid, c.threads,
               bytes / a / 1e6, bytes / b / 1e6, bytes / d / 1e6, bytes / e / 1e6);
    }
    const Cfg gs[] = {{256, 1024}, {512, 512}, {2048, 256}, {4096, 256}, {8192, 256}};
    for (const Cfg& c : gs) {
        double a = time_ms([&] { read_gridstride<4, true><<<c.grid, c.threads>>>(buf, nvec, sink); });
        double b = time_ms([&] { read_gridstride<4, false><<<c.grid, c.threads>>>(buf, nvec, sink); });
        printf("stride  grid %4d x %4d : U4 nt %6.0f | U4 plain %6.0f GB/s\n", c.grid, c.threads, bytes / a / 1e6, bytes / b / 1e6);
    }
    return 0;
}
