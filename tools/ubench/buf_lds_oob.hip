// micro-test: what does `buffer_load_dwordx4 ... offen lds` leave in LDS for lanes whose offset is past num_records?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/buf_lds_oob.hip -o /tmp/buf_lds_oob && /tmp/buf_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const unsigned *src, unsigned nbytes, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned *s32 = (unsigned *)sm;
    for (int i = threadIdx.x; i < 256; i += 64) s32[i] = 0xdeadbeefu;   // prefill
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)nbytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..63 far past the end
    const unsigned off = threadIdx.x < 32 ? threadIdx.x * 16u : 0xfffffff0u - (63 - threadIdx.x) * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)sm, 16, (int)off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = s32[i];
}
__global__ void k2(const unsigned *src, unsigned nbytes, unsigned *out, unsigned shift) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned *s32 = (unsigned *)sm;
    for (int i = threadIdx.x; i < 256; i += 64) s32[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)nbytes, 0x00020000);
    const unsigned off = threadIdx.x < 16 ? threadIdx.x * 16u + shift : 0xfffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)sm, 16, (int)off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = s32[i];
}
int main() {
    unsigned *src, *out, h[256], hs[128];
    for (int i = 0; i < 128; i++) hs[i] = 0x1000u + i;
    hipMalloc(&src, 512); hipMalloc(&out, 1024);
    hipMemcpy(src, hs, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, src, 512u, out);
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    printf("lane 0: %08x %08x  lane 31: %08x  lane 32 (OOB): %08x %08x %08x %08x  lane 63 (OOB): %08x\n", h[0], h[1], h[31 * 4], h[32 * 4], h[32 * 4 + 1], h[32 * 4 + 2], h[32 * 4 + 3], h[63 * 4]);
    int zeros = 1, kept = 1;
    for (int i = 128; i < 256; i++) { zeros &= h[i] == 0; kept &= h[i] == 0xdeadbeefu; }
    printf("OOB lanes: %s\n", zeros ? "ZERO written to LDS" : kept ? "LDS left untouched" : "something else");
    // second question: source offsets that are only 8- / 4-byte aligned
    for (unsigned sh = 4; sh <= 12; sh += 4) {
        hipLaunchKernelGGL(k2, dim3(1), dim3(64), 1024, 0, src, 512u, out, sh);
        hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int l = 0; l < 16; l++) for (int e = 0; e < 4; e++) ok &= h[l * 4 + e] == 0x1000u + (l * 16 + sh) / 4 + e;
        printf("source offset 16 l + %u: %s (lane 0 got %08x %08x %08x %08x, wanted %08x..)\n", sh, ok ? "exact" : "WRONG", h[0], h[1], h[2], h[3], 0x1000u + sh / 4);
    }
    return 0;
}
