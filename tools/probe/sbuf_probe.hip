// Probe: range-check semantics and cost of idxen+offen buffer loads on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ u4 struct_load_b128(i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4i32");

__device__ i4 mk(const void* p, int stride, unsigned nrec) {
    const unsigned long long a = (unsigned long long)p;
    i4 rs;
    rs.x = (int)(unsigned)a;
    rs.y = (int)((unsigned)(a >> 32) & 0xffff) | (stride << 16);
    rs.z = (int)nrec;
    rs.w = 0x00020000;
    return rs;
}
__global__ void k_sem(const unsigned* data, unsigned nrec, const int* idx, int n, int soff, unsigned* out) {
    i4 rs = mk(data, 512, nrec);
    for (int i = 0; i < n; ++i) {
        u4 v = struct_load_b128(rs, idx[i], (threadIdx.x % 16) * 16, soff, 0);
        if (threadIdx.x == 3) out[i] = v.x;
    }
}
__global__ void k_time(const unsigned* data, unsigned nrec, int mode, int iters, unsigned* out, long long* cyc) {
    i4 rs = mk(data, 512, nrec);
    u4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    int base = (blockIdx.x * 977 + (threadIdx.x >> 6) * 131) & 0xffff;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        int id[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int r = (base + i * 8 + k) * 4 + (lane >> 4);
            r &= 0xfffff;                     // 1M records * 512 B = 512 MB
            if (mode == 0) id[k] = r;         // valid, spread
            else if (mode == 1) id[k] = -1;   // out of range
            else if (mode == 2) id[k] = 0;    // always record 0
            else id[k] = (k == 0) ? r : -1;   // 1 valid + 7 oob
        }
        u4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = struct_load_b128(rs, id[k], (lane % 16) * 16, 0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    long long t1 = clock64();
    if (acc.x == 0x12345 && acc.y == 7) out[0] = acc.z;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    const size_t NREC = 1 << 20;
    unsigned* d; hipMalloc(&d, NREC * 512 + 4096);
    std::vector<unsigned> h(NREC * 128);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i / 128) + 1000;   // record id + 1000
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int hidx[8] = {0, 5, 99, 100, 101, -1, 1 << 21, 0x7fffffff};
    int* didx; hipMalloc(&didx, sizeof(hidx)); hipMemcpy(didx, hidx, sizeof(hidx), hipMemcpyHostToDevice);
    unsigned* dout; hipMalloc(&dout, 64 * 4);
    long long* dc; hipMalloc(&dc, 8);
    struct { const char* name; unsigned nrec; int soff; } cases[] = {
        {"nrec=100 records, soff=0", 100, 0}, {"nrec=100*512 bytes, soff=0", 100 * 512, 0},
        {"nrec=100*512 bytes, soff=512*10", 100 * 512, 5120}, {"nrec=100 records, soff=5120", 100, 5120}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d, c.nrec, didx, 8, c.soff, dout);
        unsigned ho[8]; hipMemcpy(ho, dout, 32, hipMemcpyDeviceToHost);
        printf("%s:", c.name);
        for (int i = 0; i < 8; ++i) printf(" idx=%d->%u", hidx[i], ho[i]);
        printf("\n");
    }
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_time, dim3(256), dim3(512), 0, 0, d, (unsigned)(NREC * 512), mode, 200, dout, dc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_time, dim3(256), dim3(512), 0, 0, d, (unsigned)(NREC * 512), mode, 2000, dout, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        double bytes = 256.0 * 512 * 2000 * 8 * 16;
        printf("mode %d: %.3f ms, %.1f GB/s nominal, cycles/iter(wave0)=%.1f\n", mode, ms, bytes / ms / 1e6, (double)c / 2000);
    }
    return 0;
}
