/*
 * bfstark.h -- C ABI of libbfstark_hip.so, the MI355X (gfx950) backend for the polynomial hot path of
 * aszepieniec/stark-brainfuck.  Plain pointers and sizes only; no C++ or torch types.
 *
 * The reference has no FFI layer: its boundary is the Python call surface.  Each entry point
 * below names the reference call it replaces (paths relative to /root/reference/code).  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add; stark_brainfuck_amd/ is that stub, fleshed out.
 *
 * Conventions
 *   - every function returns 0 on success or a BFS_ERR_* code; bfs_last_error() gives the message of the last
 *     failure on the calling thread.  Codes 1..5/8/9 correspond to the reference's AssertionErrors.
 *   - field elements are canonical residues mod p = 2^64 - 2^32 + 1 stored as uint64_t (little endian).
 *   - extension-field arrays are limb-major ("SoA"): c0[0..n) c1[0..n) c2[0..n), limb k at base + k*limb_stride.
 *   - pointers named d_* are device pointers on the current HIP device; h_* are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only enqueue work unless
 *     documented otherwise; the caller owns all buffers.
 */
#ifndef BFSTARK_H
#define BFSTARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    BFS_OK = 0,
    BFS_ERR_NOT_POW2 = 1,              /* ntt.py:5-6   "cannot compute ntt of non-power-of-two sequence" */
    BFS_ERR_NOT_ROOT = 2,              /* ntt.py:13-14 "primitive root must be nth root of unity" */
    BFS_ERR_NOT_PRIMITIVE = 3,         /* ntt.py:15-16 "... is not primitive nth root of unity" */
    BFS_ERR_TOO_MANY_COEFFS = 4,       /* ntt.py:166-167: more coefficients than the evaluation order */
    BFS_ERR_ZERO_IN_BATCH_INVERSE = 5, /* ntt.py:178-179 */
    BFS_ERR_BAD_ARG = 6,
    BFS_ERR_HIP = 7,
    BFS_ERR_LENGTH = 8,                /* fri.py:179-180 "initial codeword length does not match ..." */
    BFS_ERR_TOO_MANY_INDICES = 9       /* fri.py:69-70 */
};

/* ---- library / device plumbing ------------------------------------------------------------------------- */
int bfs_version(void);
const char* bfs_last_error(void);
int bfs_device_count(int* count);
int bfs_set_device(int device);
int bfs_malloc(void** d_ptr, size_t bytes);
int bfs_free(void* d_ptr);
int bfs_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int bfs_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int bfs_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
int bfs_memset(void* d_dst, int value, size_t bytes, void* stream);
int bfs_stream_synchronize(void* stream);
/* hipEvent-based timing on `stream` (bench.py: torch.cuda.Event only sees torch's own stream) */
int bfs_event_create(void** event);
int bfs_event_destroy(void* event);
int bfs_event_record(void* event, void* stream);
int bfs_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */

/* ---- field constants (host) ------------------------------------------------------------------------------ */
/* BaseField.primitive_nth_root(2^log_n)                                             algebra.py:122-136 */
uint64_t bfs_gl_primitive_root(uint32_t log_n);
/* element-wise helpers on host scalars: BaseField.multiply / inverse / __xor__     algebra.py:89-108, 39-46 */
uint64_t bfs_gl_mul(uint64_t a, uint64_t b);
uint64_t bfs_gl_inv(uint64_t a);
uint64_t bfs_gl_pow(uint64_t a, uint64_t e);

/* ---- number-theoretic transform -------------------------------------------------------------------------- */
/*
 * bfs_gl_ntt: `batch` independent length-2^log_n transforms
 *     out[b][k] = post_scale * sum_{j < n_in} in[b][j] * coset_shift^j * root^(j*k)        (natural order)
 * Replaces, depending on the arguments:
 *     ntt(root, values)                                   ntt.py:4-23     (n_in = n, coset_shift = 1, post_scale = 1)
 *     intt(root, values)                                  ntt.py:26-42    (root := root^-1, post_scale := n^-1)
 *     fast_coset_evaluate(poly, offset, generator, order) ntt.py:164-168  (n_in = len(coefficients), coset_shift = offset)
 *     Fri.Domain.evaluate / xevaluate                     fri.py:26-37    (extension field: batch = 3 limbs)
 * d_in and d_out may alias when n_in == 2^log_n and the strides agree.  Transforms b reads d_in + b*in_stride
 * (n_in elements) and writes d_out + b*out_stride (2^log_n elements).
 * Errors: BFS_ERR_NOT_ROOT / BFS_ERR_NOT_PRIMITIVE as the reference's asserts; BFS_ERR_TOO_MANY_COEFFS.
 */
int bfs_gl_ntt(const uint64_t* d_in, uint64_t n_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride,
               uint32_t log_n, uint32_t batch, uint64_t root, uint64_t coset_shift, uint64_t post_scale,
               void* stream);

/* Polynomial.scale(factor): out[b][i] = in[b][i] * factor^i                         univariate.py:168-169 */
int bfs_gl_scale(const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t stride, uint32_t batch, uint64_t factor,
                 void* stream);

/* Hadamard product (ntt.py:76) out = a * b, and batch inversion (ntt.py:177-188; Fermat inverse per element).
 * bfs_gl_batch_inverse synchronises the stream and returns BFS_ERR_ZERO_IN_BATCH_INVERSE if any input is 0. */
int bfs_gl_mul_pointwise(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n, void* stream);
int bfs_gl_batch_inverse(const uint64_t* d_in, uint64_t* d_out, uint64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BFSTARK_H */
