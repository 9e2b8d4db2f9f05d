// req_width.hip -- what does ONE random 16-byte load per lane cost at the fabric?  ("64 or 128 bytes per read request")
// N lanes each read one 16-byte word of a random 64-byte sector of a 32 GB buffer; run under
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- ./req_width          (and FETCH_SIZE, TCC_MISS_sum in passes of their own)
// RDREQ counts requests, RDREQ_32B the 32-byte ones; FETCH_SIZE counts 64-byte units... the three together give bytes per request.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void gather16(const uint4 *buf, unsigned long long n_sectors, unsigned long long *sink, unsigned long long n)
{
    unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const uint4 v = buf[(x % n_sectors) * 4]; // first 16 bytes of a random 64-byte sector
        acc += v.x + v.w;
    }
    if (acc == 0x123456789ull) *sink = acc;
}
// both 64-byte sectors of a random 128-byte line (16 bytes of each): ONE fabric request per line if a request is 128 bytes wide, TWO if 64
__global__ void gather_pair(const uint4 *buf, unsigned long long n_lines, unsigned long long *sink, unsigned long long n)
{
    unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const uint4 *l = buf + (x % n_lines) * 8;
        const uint4 v = l[0], w = l[4];
        acc += v.x + w.w;
    }
    if (acc == 0x123456789ull) *sink = acc;
}
__global__ void scatter16(uint4 *buf, unsigned long long n_sectors, unsigned long long n)
{
    unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    for (; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        buf[(x % n_sectors) * 4] = make_uint4((unsigned)i, 1, 2, 3);
    }
}
int main(int argc, char **argv)
{
    const unsigned long long bytes = (argc > 1 ? strtoull(argv[1], 0, 0) : 32ull) << 30, n = argc > 2 ? strtoull(argv[2], 0, 0) : 1ull << 28;
    uint4 *buf; unsigned long long *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        gather16<<<256 * 16, 256>>>(buf, bytes / 64, sink, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("gather16: %llu random 16-byte loads (one per 64-byte sector) of %llu GB in %.2f ms = %.2f G/s; %.2f TB/s at 64 B each, %.2f at 128 B\n", n, bytes >> 30, ms, n / ms / 1e6, n * 64 / ms / 1e9, n * 128 / ms / 1e9);
        hipEventRecord(a);
        gather_pair<<<256 * 16, 256>>>(buf, bytes / 128, sink, n / 2);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("gather_pair: %llu random 128-byte lines, 16 bytes of either sector, in %.2f ms = %.2f G lines/s\n", n / 2, ms, n / 2 / ms / 1e6);
        hipEventRecord(a);
        scatter16<<<256 * 16, 256>>>(buf, bytes / 64, n);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("scatter16: %llu random 16-byte stores in %.2f ms = %.2f G/s\n", n, ms, n / ms / 1e6);
    }
    return 0;
}
