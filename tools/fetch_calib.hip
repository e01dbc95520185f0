// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access patterns this repo uses
// (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream; other widths are
// "uncalibrated -- calibrate on a known byte count in your own access pattern").
// Reads a 512 MiB buffer (larger than L2 + Infinity Cache) exactly once with
//   k_ubyte  : global_load_ubyte, 64 consecutive bytes per wave instruction (the SSIM loader's pattern)
//   k_dwordx4: 16 B per lane, 1 KiB per wave instruction
// run under: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_ubyte(const uint8_t* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += p[i];
    if (acc == 0xFFFFFFFFu) *out = acc;
}

__global__ void k_dwordx4(const uint4* __restrict__ p, size_t n16, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0xFFFFFFFFu) *out = acc;
}

int main()
{
    const size_t n = 512ull << 20;
    uint8_t* d; unsigned* o;
    CHECK(hipMalloc(&d, n)); CHECK(hipMalloc(&o, 4));
    CHECK(hipMemset(d, 1, n));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_ubyte, dim3(256 * 16), dim3(256), 0, 0, d, n, o);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_dwordx4, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)d, n / 16, o);
        CHECK(hipDeviceSynchronize());
    }
    printf("read %zu bytes per kernel\n", n);
    return 0;
}
