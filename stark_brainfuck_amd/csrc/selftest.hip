// selftest.hip -- device-versus-host check of the field primitives and of compositions of them (constants included).
// The same inline functions (gl.hpp, ntt_core.hpp) compile for both sides; the host side is plain 128-bit arithmetic.
// It exists because hipcc (ROCm 7.2) miscompiled gl_add(gl_sub(a, b), 1) -- see gl_sub in gl.hpp -- and nothing but a
// comparison on the device can notice that kind of error.  Called by tests/test_gpu_parity.py.
#define BFS_GL_SUB4       // every operation below runs the four-instruction subtraction (gl.hpp); operations 34.. the other form
#include "runtime.hpp"

namespace bfs {

constexpr int ST_OPS = 50;

BFS_HD void selftest_ops(u64 a, u64 b, u64* o) {
    const u64 d = gl_sub(a, b), s = gl_add(a, b), m = gl_mul(a, b);
    o[0] = s; o[1] = d; o[2] = m; o[3] = gl_neg(a);
    o[4] = gl_add(a, 1ULL); o[5] = gl_sub(a, 1ULL); o[6] = gl_sub(a, 2ULL); o[7] = gl_sub(d, 2ULL);
    o[8] = gl_add(d, 1ULL); o[9] = gl_mul(a, gl_sub(gl_sub(b, a), 2ULL));
    o[10] = gl_add(s, 1ULL); o[11] = gl_add(m, 1ULL); o[12] = gl_sub(m, 1ULL); o[13] = gl_add(d, GL_P - 1);
    o[14] = gl_add(gl_add(d, s), m); o[15] = gl_sub(gl_add(m, 7ULL), d); o[16] = gl_mul(gl_add(d, 1ULL), gl_sub(s, 1ULL));
    o[17] = gl_add(gl_neg(d), 1ULL); o[18] = gl_sub(1ULL, d); o[19] = gl_sub(0ULL, m); o[20] = gl_add(1ULL, gl_mul(d, 44ULL));
    o[21] = gl_mul(gl_mul_lazy(gl_mul_lazy(a, b), gl_add(d, 1ULL)), 1ULL);
    o[22] = gl_add(gl_sub(gl_sub(a, b), b), 1ULL); o[23] = gl_sub(gl_add(gl_add(a, 1ULL), 1ULL), b);
    o[24] = mul_pow2<12>(d); o[25] = mul_pow2<36>(s); o[26] = mul_pow2<48>(d); o[27] = mul_pow2<72>(m); o[28] = mul_pow2<84>(gl_add(d, 1ULL));
    const Xfe x{{a, b, d}}, y{{s, m, a}};
    const Xfe z = xfe_mul(xfe_sub_base(xfe_add(x, y), 1ULL), xfe_base_sub(2ULL, xfe_scale(y, b)));
    o[29] = z.c[0]; o[30] = z.c[1]; o[31] = z.c[2];
    o[32] = gl_inv(a); o[33] = gl_mul(gl_inv(d), d);          // the addition chain with non-canonical intermediate squares
    // both instruction sequences of the subtraction and of the reduction's first step (gl_sub4 / gl_sub5, gl_reduce128_t<.., SUB4>)
    o[34] = gl_sub5(a, b); o[35] = gl_sub5(gl_sub5(b, a), 2ULL); o[36] = gl_add(gl_sub5(a, b), 1ULL); o[37] = gl_sub4(gl_sub5(m, s), d);
    o[38] = gl_sub4(b, a); o[39] = gl_add(gl_sub4(gl_sub4(a, b), b), 1ULL); o[40] = gl_mul_other_form(a, b); o[41] = gl_mul_other_form(gl_mul_other_form(d, s), m);
    // the unreduced sum (gl_add_lazy: any 64-bit first operand, canonical second) and everything that may consume it
    const u64 l = gl_add_lazy(a, b), any = ~a;
    o[42] = l; o[43] = gl_add_lazy(l, d); o[44] = gl_sub(l, d); o[45] = mul_pow2<48>(l); o[46] = gl_mul(l, m);
    o[47] = gl_add_lazy(any, b); o[48] = gl_canon(gl_sub(any, b)); o[49] = mul_pow2<12>(gl_add_lazy(gl_add_lazy(any, b), s));
}

__global__ void selftest_kernel(const u64* in, u64* out, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) selftest_ops(in[2 * i], in[2 * i + 1], out + ST_OPS * i);
}

}  // namespace bfs

using namespace bfs;

extern "C" int bfs_selftest_field(uint32_t log_count, uint64_t* mismatches) {
    const u64 n = 1ull << log_count;
    std::vector<u64> in(2 * n), got(ST_OPS * n);
    const u64 edges[] = {0, 1, 2, 3, 0xFFFFFFFFULL, 0x100000000ULL, 0x100000001ULL, GL_P - 1, GL_P - 2, GL_P - 3, 0xFFFFFFFF00000000ULL - 1,
                         0xFFFFFFFEFFFFFFFFULL, 300, 44, 0xFFFFFFFE00000001ULL, 0x8000000000000000ULL,
                         // products with 2 or 3 that fit 64 bits but are >= p: the reduction's value is only non-canonical in its last step
                         // (one product in 2^32 lands there by chance)
                         0x5555555555555555ULL, 0x7FFFFFFF80000001ULL, 0x7FFFFFFFFFFFFFFFULL, 0x55555555AAAAAAABULL,
                         // ... and operands whose 96-bit shifts (mul_pow2 with K < 32) do the same
                         0x000FFFFFFFF00001ULL, 0x000FFFFFFFFFFFFFULL};
    const u64 ne = sizeof(edges) / sizeof(edges[0]);
    u64 s = 3;
    auto rnd = [&]() { s += 0x9E3779B97F4A7C15ULL; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return (z ^ (z >> 31)) % GL_P; };
    for (u64 i = 0; i < n; ++i) {
        if (i < ne * ne) { in[2 * i] = edges[i / ne]; in[2 * i + 1] = edges[i % ne]; }
        else { in[2 * i] = (i & 64) ? rnd() % 300 : ((i & 128) ? GL_P - 1 - rnd() % 300 : rnd()); in[2 * i + 1] = (i & 32) ? rnd() % 300 : rnd(); }
    }
    u64 *d_in = nullptr, *d_out = nullptr;
    BFS_HIP(hipMalloc(&d_in, in.size() * sizeof(u64)));
    BFS_HIP(hipMalloc(&d_out, got.size() * sizeof(u64)));
    BFS_HIP(hipMemcpy(d_in, in.data(), in.size() * sizeof(u64), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, 0, d_in, d_out, n);
    BFS_HIP(hipGetLastError());
    BFS_HIP(hipMemcpy(got.data(), d_out, got.size() * sizeof(u64), hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_out);
    u64 bad = 0;
    for (u64 i = 0; i < n; ++i) {
        u64 want[ST_OPS];
        selftest_ops(in[2 * i], in[2 * i + 1], want);
        for (int j = 0; j < ST_OPS; ++j)
            if (want[j] != got[ST_OPS * i + j]) {
                if (!bad) set_error("field self-test: operation %d on a=%llx b=%llx gives %llx on the device, %llx on the host", j,
                                    (unsigned long long)in[2 * i], (unsigned long long)in[2 * i + 1], (unsigned long long)got[ST_OPS * i + j], (unsigned long long)want[j]);
                ++bad;
            }
    }
    *mismatches = bad;
    return BFS_OK;
}
