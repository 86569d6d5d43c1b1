// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access width
// (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of a wide coalesced streaming read (16 B/lane) ... other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Measurement tool only (tools/pmc/calibrate.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`); not product code.
//   build: hipcc --offload-arch=gfx950 -O3 tools/pmc/pmc_calib.hip -o tools/pmc/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T> struct Acc;
template <> struct Acc<float>  { static __device__ float f(float v)  { return v; } };
template <> struct Acc<float2> { static __device__ float f(float2 v) { return v.x + v.y; } };
template <> struct Acc<float4> { static __device__ float f(float4 v) { return v.x + v.y + v.z + v.w; } };
template <> struct Acc<unsigned short> { static __device__ float f(unsigned short v) { return (float)v; } };

// streaming read of n elements of T, one element per lane per iteration (coalesced: a wave covers 64 * sizeof(T) contiguous bytes)
template <typename T> __global__ void rd_kernel(const T* __restrict__ src, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += Acc<T>::f(src[i]);
    if (s == 123.456f) out[0] = s;
}
template <typename T> __device__ T fillv(float v);
template <> __device__ float fillv<float>(float v) { return v; }
template <> __device__ float2 fillv<float2>(float v) { return make_float2(v, v); }
template <> __device__ float4 fillv<float4>(float v) { return make_float4(v, v, v, v); }
template <> __device__ unsigned short fillv<unsigned short>(float v) { return (unsigned short)v; }
template <typename T> __global__ void wr_kernel(T* __restrict__ dst, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = fillv<T>(v);
}
// 4 B/lane gather with a row stride: lane l of a wave reads element (row * pitch + l * stride) -- stride 2 touches every other float
// (the bicubic sampler's pattern at down-scale 2: half of every 128-byte line is used)
__global__ void rd_stride_kernel(const float* __restrict__ src, size_t n, int stride, float* out) {
    float s = 0.f;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; i < n; i += (size_t)gridDim.x * blockDim.x * stride) s += src[i];
    if (s == 123.456f) out[0] = s;
}

int main() {
    const size_t bytes = (size_t)1 << 30;      // 1 GiB: four times the 256 MiB Infinity Cache
    void *a, *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&o, 4096) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(a, 1, bytes);
    (void)hipDeviceSynchronize();
    const int G = 256 * 8, B = 256;
    for (int rep = 0; rep < 2; ++rep) {
        rd_kernel<unsigned short><<<G, B>>>((const unsigned short*)a, bytes / 2, (float*)o);
        rd_kernel<float><<<G, B>>>((const float*)a, bytes / 4, (float*)o);
        rd_kernel<float2><<<G, B>>>((const float2*)a, bytes / 8, (float*)o);
        rd_kernel<float4><<<G, B>>>((const float4*)a, bytes / 16, (float*)o);
        rd_stride_kernel<<<G, B>>>((const float*)a, bytes / 4, 2, (float*)o);
        wr_kernel<unsigned short><<<G, B>>>((unsigned short*)a, bytes / 2, 1.f);
        wr_kernel<float><<<G, B>>>((float*)a, bytes / 4, 1.f);
        wr_kernel<float2><<<G, B>>>((float2*)a, bytes / 8, 1.f);
        wr_kernel<float4><<<G, B>>>((float4*)a, bytes / 16, 1.f);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("pmc_calib: %s; every kernel touches %zu bytes (stride-2 read: every 128-byte line of them, half of each line used)\n", hipGetErrorString(e), bytes);
    return e != hipSuccess;
}
