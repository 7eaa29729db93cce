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
/* HBM comes from a pool inside the library (size-class free lists over hipMalloc: the driver calls cost 50-500 us and
 * hipFree synchronises the device).  bfs_malloc / bfs_free keep hipMalloc / hipFree semantics (bfs_free waits for the
 * device).  The *_async pair is stream-ordered like hipMallocAsync / hipFreeAsync: the block goes back to the pool at
 * once, work already queued on `stream` may still use it, and a later request on the same stream gets it without
 * waiting (a request on another stream synchronises `stream` first).  bfs_pool_trim returns the cached blocks to the
 * driver; BFS_POOL=0 in the environment disables caching.  bfs_host_alloc hands out pinned host memory from a pool of
 * the same kind (staging buffers for bfs_memcpy_h2d: pageable sources copy at ~1 GB/s, pinned ones at link speed). */
int bfs_malloc(void** d_ptr, size_t bytes);
int bfs_free(void* d_ptr);
int bfs_malloc_async(void** d_ptr, size_t bytes, void* stream);
int bfs_free_async(void* d_ptr, void* stream);
int bfs_pool_trim(void);
int bfs_pool_stats(size_t* live_bytes, size_t* cached_bytes);
int bfs_host_alloc(void** h_ptr, size_t bytes);
int bfs_host_free(void* h_ptr);
int bfs_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int bfs_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int bfs_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
int bfs_memset(void* d_dst, int value, size_t bytes, void* stream);
int bfs_stream_synchronize(void* stream);
/* a HIP stream of the current device for callers without a HIP runtime of their own (a foreign hipStream_t / torch stream handle is
 * accepted wherever `stream` is taken; 0 is the default stream).  Independent provers on different streams may run concurrently,
 * also from different host threads. */
int bfs_stream_create(void** stream);
int bfs_stream_destroy(void* stream);
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
 * Transform b reads d_in + b*in_stride (n_in elements) and writes d_out + b*out_stride (2^log_n elements).  The input is left
 * untouched when input and output do not overlap, and then the transform needs no intermediate memory (its first pass writes the
 * output, later passes run in place there); the two ranges may also overlap in any way -- d_in == d_out, an output that starts inside
 * the input -- in which case the first two passes go through a library buffer of the output's size.  Stream-ordered: the call only
 * enqueues kernels (it never measures, synchronises or allocates candidate buffers by itself; it may be captured into a hipGraph once
 * its tables exist, i.e. after one plain call of the same shape).  A pair of buffers that bfs_ntt_tune() found a faster route for is
 * served through the library buffer that call kept.
 * Errors: BFS_ERR_NOT_ROOT / BFS_ERR_NOT_PRIMITIVE as the reference's asserts; BFS_ERR_TOO_MANY_COEFFS.
 */
int bfs_gl_ntt(const uint64_t* d_in, uint64_t n_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride,
               uint32_t log_n, uint32_t batch, uint64_t root, uint64_t coset_shift, uint64_t post_scale,
               void* stream);

/*
 * bfs_ntt_tune (no reference counterpart; optional): for a LARGE out-of-place transform (>= 256 MiB, all n inputs read, several
 * passes) that the caller is going to repeat on the same (d_in, d_out) pair -- a benchmark step, a prover's pooled buffers --, choose by
 * measurement where the first pass writes: straight into the output or through one of three library buffers of the output's size
 * (how fast the first, transposing pass streams depends on the physical placement of the PAIR of buffers: 405-510 us for the same
 * launch between different pairs).  What the call does, so that nobody is surprised by it: it allocates three buffers of
 * the output's size (only if four such sizes are free, otherwise the pair stays direct), runs passes 0 + 1 of the transform 18 x 4
 * times (~63 ms at 8 x 2^24; d_out ends up holding an unfinished transform), synchronises `stream` once, frees the buffers that lost and
 * KEEPS the winner, if any, for as long as the pair is remembered (*route: -1 direct -- nothing kept --, 0..2 that buffer).  Not to be
 * called on a stream that is being captured (BFS_ERR_BAD_ARG).  Transforms too small to matter, overlapping buffers and
 * BFS_NTT_WS_PROBE=0 make it a no-op.  The pair is forgotten -- and its buffer released at the next bfs_pool_trim() / bfs_ntt_tune() --
 * when either buffer goes back to the library's pool (bfs_free, bfs_free_async); for memory the library does not own, call
 * bfs_ntt_route_forget(ptr) before freeing it (NULL: forget every pair), which synchronises the device and frees what is no longer
 * needed at once.  *forgotten (optional): how many pairs went.
 */
int bfs_ntt_tune(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride, uint32_t log_n, uint32_t batch, uint64_t root,
                 void* stream, int* route);
int bfs_ntt_route_forget(const void* d_ptr, size_t* forgotten);

/* Diagnostics of the route measurement (no reference counterpart): what the LAST measurement of this process read.
 * us[0] = passes 0 + 1 straight into the output, us[1..3] = through library buffer 0..2 (microseconds); *route = -1 direct or the
 * buffer chosen; *probes = measurements taken so far (0: none yet, us / route are then 0 / -1).  Any pointer may be NULL. */
int bfs_ntt_route_probe_info(float* us, int* route, unsigned long long* probes);

/* Polynomial.scale(factor): out[b][i] = in[b][i] * factor^i                         univariate.py:168-169 */
int bfs_gl_scale(const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t stride, uint32_t batch, uint64_t factor,
                 void* stream);

/* Hadamard product (ntt.py:76) out = a * b, and batch inversion (ntt.py:177-188; Fermat inverse per element).
 * bfs_gl_batch_inverse synchronises the stream and returns BFS_ERR_ZERO_IN_BATCH_INVERSE if any input is 0. */
int bfs_gl_mul_pointwise(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n, void* stream);
int bfs_gl_batch_inverse(const uint64_t* d_in, uint64_t* d_out, uint64_t n, void* stream);
/* The same two over the cubic extension F_p[X]/(X^3 - X + 1) (limb planes `*_stride` words apart): what fast_multiply's
 * hadamard_product (ntt.py:74-76) and fast_coset_divide's batch_inverse (ntt.py:226-229) compute when their operands are
 * ExtensionFieldElements -- the way Table.ldex reaches them (table.py:133-134 -> ntt.py:126-161 -> 82-98 -> 45-79), with
 * ExtensionField.multiply / inverse (extension_field.py:71-83) per point.  In place (d_out == an input) is allowed.
 * bfs_xfe_batch_inverse synchronises the stream; a zero element gives BFS_ERR_ZERO_IN_BATCH_INVERSE (its output is zero). */
int bfs_xfe_mul_pointwise(const uint64_t* d_a, uint64_t a_stride, const uint64_t* d_b, uint64_t b_stride, uint64_t* d_out, uint64_t out_stride,
                          uint64_t n, void* stream);
int bfs_xfe_batch_inverse(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride, uint64_t n, void* stream);

/* ---- proof stream / Fiat-Shamir (host) -------------------------------------------------------------------- */
/*
 * ProofStream of the reference (ip.py:4-30): a list of Python objects, serialised with pickle.dumps and hashed
 * with SHAKE256.  Objects are built through handles (opaque, valid for the lifetime of the stream); building the
 * same handle into several containers reproduces Python's shared-object memoisation.
 *   bfs_ps_obj_xfe  : ExtensionFieldElement whose coefficients live in the xfield's own BaseField (extension_field.py:88-98)
 *   bfs_ps_obj_bfe  : BaseFieldElement pointing at BaseField instance `field_id`: 1 = the instance inside the xfield's
 *                     modulus, 0 = a second BaseField.main() instance, 2.. = further instances (the reference creates one
 *                     per class that calls BaseField.main(): vm.py:70, brainfuck_stark.py:21; pickle writes each out once)
 *   bfs_ps_obj_xfe_from : ExtensionFieldElement over explicit coefficient objects (handles of bfs_ps_obj_bfe, leading one
 *                     non-zero): for elements that share BaseFieldElement objects with other elements or whose coefficients
 *                     point at a foreign BaseField instance -- both happen in the reference (univariate.py:23-27 returns the
 *                     other operand's polynomial when one summand is zero)
 *   bfs_ps_push                ProofStream.push                ip.py:9-10
 *   bfs_ps_serialize(count)    pickle.dumps(objects[:count])   ip.py:18-19,24-25 (count >= #objects: the whole stream);
 *                              two-call pattern: pass out = NULL to query *length
 *   bfs_ps_fiat_shamir(count)  shake_256(serialize).digest(n)  ip.py:21-25
 */
void* bfs_ps_new(void);
void bfs_ps_free(void* ps);
/* ProofStream.deserialize (ip.py:27-30: pickle.loads of the proof bytes) straight into a native stream: the objects of the pickled list,
 * object identities included, with handles 1..n in stream order.  NULL (see bfs_last_error) unless serialising the result reproduces the
 * input byte for byte -- then the caller unpickles with CPython and rebuilds through bfs_ps_obj_* as before. */
void* bfs_ps_loads(const uint8_t* data, size_t len);
uint64_t bfs_ps_obj_bytes(void* ps, const uint8_t* data, size_t len);
uint64_t bfs_ps_obj_int(void* ps, uint64_t value);
uint64_t bfs_ps_obj_xfe(void* ps, const uint64_t limbs[3]);
uint64_t bfs_ps_obj_bfe(void* ps, uint64_t value, int field_id);
uint64_t bfs_ps_obj_xfe_from(void* ps, const uint64_t* coefficient_handles, size_t n);
uint64_t bfs_ps_obj_list(void* ps, const uint64_t* handles, size_t n);
uint64_t bfs_ps_obj_tuple(void* ps, const uint64_t* handles, size_t n);
int bfs_ps_push(void* ps, uint64_t handle);
size_t bfs_ps_num_objects(void* ps);
uint64_t bfs_ps_object_at(void* ps, size_t index);
int bfs_ps_serialize(void* ps, size_t count, uint8_t* out, size_t capacity, size_t* length);
int bfs_ps_fiat_shamir(void* ps, size_t count, uint8_t* out, size_t num_bytes);
/* ProofStream.verifier_fiat_shamir (ip.py:27-30) ahead of time: the hashes over objects[:counts[i]] of a stream made by bfs_ps_loads
 * are offered to the library's helper threads; a later bfs_ps_fiat_shamir(ps, counts[i], ..., num_bytes) picks the result up (or
 * computes it itself when no helper has got to it).  Returns how many were offered (0 without helper threads: not an error). */
size_t bfs_ps_prefetch_fiat_shamir(void* ps, const size_t* counts, size_t n, size_t num_bytes);
/* Merkle.verify / SaltedMerkle.verify (merkle.py:54-63, salted_merkle.py:55-68) on objects of this stream, natively: leaf =
 * blake2b(pickle.dumps(element) [+ pickle.dumps(salt)]; salt_handle 0 = unsalted), then the path (a list of byte strings) folded by the
 * parity of `index`; *ok = 1 when the result equals root.  bfs_xfe_inner_product: sum of weights[i] * terms[i] over the extension field
 * (3 limbs each), the verifier's inner product (brainfuck_stark.py:553-554). */
int bfs_ps_merkle_verify(void* ps, uint64_t element_handle, uint64_t salt_handle, uint64_t path_handle, uint64_t index, const uint8_t* root,
                         size_t root_len, int* ok);
int bfs_xfe_inner_product(const uint64_t* weights, const uint64_t* terms, size_t count, uint64_t out[3]);
/* push(bytes(digest)) followed by fiat_shamir over everything, computed the way bfs_fri_commit overlaps it with a tree kernel: the
 * SHAKE256 blocks in front of the digest's payload are absorbed before the digest is known (same result as the two calls). */
int bfs_ps_push_digest_fiat_shamir(void* ps, const uint8_t digest[64], uint8_t* out, size_t num_bytes);
/* the same for `count` digests in a row (digests: count * 64 bytes; out: count * num_bytes bytes, the Fiat-Shamir bytes after each
 * push), computed the way bfs_fri_commit does behind a long transcript: every challenge hashes the whole stream again under a new
 * frame header (ip.py:21-25), so the pickles of all `count` coming streams are laid out up front and helper threads absorb their
 * prefixes side by side (BFS_HELPER_THREADS, default 4; 0 = off).  *used_lookahead (optional): 0 when it fell back to one at a time
 * (fewer than two objects in the stream, or no helper threads). */
int bfs_ps_push_digests_fiat_shamir(void* ps, const uint8_t* digests, size_t count, uint8_t* out, size_t num_bytes, int* used_lookahead);
/* pickle.dumps(obj) of one object on its own (leaf preimages: merkle.py:30, salted_merkle.py:32-33); two-call pattern */
int bfs_ps_obj_dumps(void* ps, uint64_t handle, uint8_t* out, size_t capacity, size_t* length);
/* introspection (to hand objects created by bfs_fri_prove back to the host language):
 * kind: 0 bytes, 1 int, 3 list, 4 tuple, 100 extension element, 101 base element */
int bfs_ps_obj_kind(void* ps, uint64_t handle);
size_t bfs_ps_obj_len(void* ps, uint64_t handle);
uint64_t bfs_ps_obj_item(void* ps, uint64_t handle, size_t i);
int bfs_ps_obj_get_bytes(void* ps, uint64_t handle, uint8_t* out, size_t capacity);
int bfs_ps_obj_get_limbs(void* ps, uint64_t handle, uint64_t limbs[3]);
/* BaseField.sample / ExtensionField.sample: big-endian bytes -> element        algebra.py:138-142, extension_field.py:100-111 */
/* BrainfuckStark.sample_weights (brainfuck_stark.py:104-112): `count` extension elements, weight i = ExtensionField.sample of
 * blake2b(randomness || i zero bytes); out: 3 * count limbs.  Host code (a proof draws ~170 of them). */
int bfs_sample_weights(const uint8_t* randomness, size_t len, size_t count, uint64_t* out);
uint64_t bfs_gl_sample(const uint8_t* bytes, size_t len);
void bfs_xfe_sample(const uint8_t* bytes, size_t len, uint64_t out[3]);

/* Scattered reads from HBM in one round trip: request r delivers `nwords` words d_base[0], d_base[stride], d_base[2*stride], ...;
 * results are packed into h_out in request order.  One small kernel writes into pinned host memory (no copy commands); used for
 * the handful of codeword rows and tree nodes a proof reveals (fri.py:141-176, brainfuck_stark.py:315-333).  `out_offset` is
 * ignored on input.  Synchronises the stream. */
typedef struct bfs_gather_request {
    const uint64_t* d_base;
    uint32_t nwords, stride;
    uint64_t out_offset;
} bfs_gather_request;
int bfs_gather(const bfs_gather_request* requests, uint32_t count, uint64_t* h_out, void* stream);
/*
 * bfs_stark_push_openings   the openings of BrainfuckStark.prove (brainfuck_stark.py:315-333) pushed onto the proof stream natively:
 *     for every index and every distance (the first distance is 0, then the tables' unit distances): base row at index + distance,
 *     its (salt, path), extension row, its (salt, path); then per index the combination leaf and its path.
 *     base_row / ext_row: the gather requests of row 0 (d_base is advanced by the row index): the base row's first three words are
 *     an ExtensionFieldElement (the randomizer codeword), every further word a BaseFieldElement of BaseField instance base_field_id; the
 *     extension row holds n_ext_cols elements of three words, ext_moduli[c] != 0 meaning that the rows i = i' mod ext_moduli[c] share
 *     their coefficient objects in column c.  Trees: 2 n digests each (merkle.py layout); salts: 24 bytes per leaf, in HBM
 *     (*_salts_on_device != 0) or on the host.  out_leaf_handles[a] = handle of the combination leaf of indices[a] (for
 *     bfs_fri_session_alias).  One gather for everything; synchronises the stream.
 */
int bfs_stark_push_openings(void* ps, const bfs_gather_request* base_row, uint32_t n_base_req, int32_t base_field_id,
                            const bfs_gather_request* ext_row, uint32_t n_ext_req, const uint64_t* ext_moduli, uint32_t n_ext_cols,
                            uint64_t n, const uint8_t* d_base_nodes, const uint8_t* base_salts, int base_salts_on_device,
                            const uint8_t* d_ext_nodes, const uint8_t* ext_salts, int ext_salts_on_device,
                            const uint64_t* d_combination, uint64_t combination_stride, const uint8_t* d_combination_nodes,
                            const uint64_t* indices, uint32_t n_indices, const uint64_t* distances, uint32_t n_distances,
                            uint64_t* out_leaf_handles, void* stream);

/* ---- Merkle trees ------------------------------------------------------------------------------------------- */
/*
 * Tree layout = the reference's `nodes` list (merkle.py:26-44): 2*npo2 digests of 64 bytes, npo2 = next power of
 * two >= n, leaf i at index npo2 + i, parent k = BLAKE2b-512(nodes[2k] || nodes[2k+1]), root at index 1.  Index 0
 * and the slots of absent leaves are not written (the reference keeps 32 zero bytes there; the parent of an absent
 * leaf hashes those 32 zero bytes, which these kernels reproduce).  d_nodes must hold 2*npo2*64 bytes and be 16-byte aligned
 * (BFS_ERR_BAD_ARG otherwise).
 *   bfs_merkle_build_xfe   Merkle(codeword) over ExtensionFieldElement leaves   merkle.py:8-41 (leaf = blake2b(pickle.dumps(e)))
 *   bfs_merkle_build_bfe   same over BaseFieldElement leaves (stand-alone BaseField instance)
 *   bfs_merkle_build_bytes same over caller-pickled leaves: message i = lengths[i] bytes at d_data + 8*word_offsets[i]
 *                          (arbitrary picklable leaves, merkle.py:30; salted leaves, salted_merkle.py:32-35)
 *   bfs_merkle_open        Merkle.open(index): `depth` sibling digests, leaf level first      merkle.py:46-52 (synchronous)
 */
int bfs_merkle_build_xfe(const uint64_t* d_limbs, uint64_t limb_stride, uint64_t n, uint8_t* d_nodes, void* stream);
int bfs_merkle_build_bfe(const uint64_t* d_values, uint64_t n, uint8_t* d_nodes, void* stream);
int bfs_merkle_build_bytes(const uint8_t* d_data, const uint64_t* d_word_offsets, const uint32_t* d_lengths, uint64_t n,
                           uint8_t* d_nodes, void* stream);
int bfs_merkle_open(const uint8_t* d_nodes, uint32_t depth, uint64_t index, uint8_t* h_path, void* stream);
/*
 * bfs_merkle_build_rows   SaltedMerkle(list(zip(*codewords)))  (brainfuck_stark.py:178-179, 197-198; salted_merkle.py:22-47):
 *     leaf i = blake2b(pickle.dumps(tuple of the i-th element of every column) || pickle.dumps(salt_i)).  Columns are
 *     codewords in HBM (at most 32, at most 16 of them extension columns): an extension column is three limb planes of n
 *     words (elements of the xfield's own BaseField), a base column n words whose elements point at BaseField instance
 *     `field_id` (as in bfs_ps_obj_bfe).  salts: n x 24 bytes, on the host or (salts_on_device != 0) in HBM; NULL for unsalted tuples.  The pickle of every
 *     row is synthesised on the GPU and streamed into BLAKE2b (csrc/rows.hip); synchronises the stream.  n must be a power
 *     of two for a SaltedMerkle (salted_merkle.py:22).
 */
typedef struct bfs_row_column {
    const uint64_t* d_values;
    int32_t is_ext;
    int32_t field_id;
} bfs_row_column;
/* bfs_merkle_build_rows_range: the same over a RANGE of n rows of longer columns: d_values point at the first row of the range, the
 * limb planes of an extension column are limb_stride words apart (the full column length), salts are those of the range.  The tree
 * it writes is the subtree over these rows: a rank of a row-sharded commitment hashes its rows with this (stark_brainfuck_amd/shard.py). */
int bfs_merkle_build_rows_range(const bfs_row_column* columns, uint32_t ncols, uint64_t n, uint64_t limb_stride, const uint8_t* salts,
                                int salts_on_device, uint8_t* d_nodes, void* stream);
int bfs_merkle_build_rows(const bfs_row_column* columns, uint32_t ncols, uint64_t n, const uint8_t* salts, int salts_on_device,
                          uint8_t* d_nodes, void* stream);
/* bfs_merkle_build_rows_range that also hands back the root (SaltedMerkle.root(), salted_merkle.py:51-52: what the prover pushes
 * next, brainfuck_stark.py:179, 198) in the read-back the call ends with anyway.  h_root may be NULL. */
int bfs_merkle_build_rows_root(const bfs_row_column* columns, uint32_t ncols, uint64_t n, uint64_t limb_stride, const uint8_t* salts,
                               int salts_on_device, uint8_t* d_nodes, uint8_t h_root[64], void* stream);
/* The flattened template of one row pattern of a column layout (csrc/rows.hip; `code` = 2 bits per extension column, how many
 * coefficients its element stores; d_values are not looked at): what tools/gen_rows.py unrolls into csrc/rows_generated.hpp and what
 * the tests compare that header with.  out_header = {steps, integers, constant bytes of the tuple pickle, bytes of the salt pickle};
 * a step is two words (kind | a << 32, data: csrc/rows_core.hpp RowStep); out_ints[k] = column | limb << 8 of the k-th integer of
 * the row; *out_hash = the number a generated kernel is matched by.  Host only, no GPU work. */
int bfs_row_template_steps(const bfs_row_column* columns, uint32_t ncols, uint32_t code, int salted, uint32_t out_header[4],
                           uint64_t* out_steps, uint32_t steps_cap, uint32_t* out_ints, uint32_t ints_cap, uint64_t* out_hash);
/* How many leaf launches of this process went through a kernel of csrc/rows_generated.hpp (the tests check that the prover's two
 * layouts do and that other layouts do not). */
uint64_t bfs_row_generated_launches(void);
/* nwords (a multiple of 8) pseudo-random words in HBM: 64-byte block j = BLAKE2b-512(seed || j).  For salts that never visit
 * the host (the reference draws os.urandom(24) per leaf, salted_merkle.py:25; the caller seeds this from os.urandom(32)). */
int bfs_random_fill(const uint8_t seed[32], uint64_t* d_out, uint64_t nwords, void* stream);
/* `count` pseudo-random extension elements (limb planes `limb_stride` words apart) in HBM: ExtensionField.sample
 * (extension_field.py:100-111) of 27 bytes per element, element i taking bytes [27 i, 27 i + 27) of the stream made of the first 63
 * bytes of every block BLAKE2b-512(seed || b), b = 0, 1, ... (limb j = the j-th run of 9 bytes as a big-endian integer mod p).  For
 * the randomizer polynomial of brainfuck_stark.py:162-165 without the host in the loop. */
int bfs_xfe_sample_fill(const uint8_t seed[32], uint64_t* d_out, uint64_t count, uint64_t limb_stride, void* stream);

/* ---- trace tables: padding on the device -------------------------------------------------------------------- */
/*
 * bfs_trace_pad: Table.pad of every trace table (table.py:25 with processor_table.py:24-35, instruction_table.py:19-25,
 * memory_table.py:40-44, io_table.py:17-21) on rows that are already in HBM, ROW-major as the virtual machine wrote them
 * (rows x row_stride words, the first `width` of a row are used): d_out receives the padded table COLUMN-major (width x height
 * words, residues reduced mod p), the masks one byte per row for the scans of the table's extension:
 *   kind 0 processor   (width >= 7): d_mask0 = current instruction != 0, d_mask1 = it is ',', d_mask2 = it is '.'
 *   kind 1 instruction (width >= 2): d_mask0 = rows of the running product, d_mask1 = rows of the running evaluation
 *   kind 2 memory      (width >= 4): d_mask0 = non-dummy rows
 *   kind 3 / 4 input / output: padding rows are zero rows, no masks.
 * Stream-ordered; at most five tables per call.
 */
typedef struct bfs_trace_pad_table {
    const uint64_t* d_rows;
    uint64_t rows, row_stride, height;
    uint64_t* d_out;
    uint8_t *d_mask0, *d_mask1, *d_mask2;
    int32_t kind;
    uint32_t width;
} bfs_trace_pad_table;
int bfs_trace_pad(const bfs_trace_pad_table* tables, uint32_t count, void* stream);

/* ---- FRI ---------------------------------------------------------------------------------------------------- */
/*
 * bfs_xfe_fold: one split-and-fold round (fri.py:127-128) of a limb-major extension codeword of length 2^log_n:
 *     out[i] = 2^-1 * ((1 + alpha/(offset*omega^i)) * in[i] + (1 - alpha/(offset*omega^i)) * in[n/2 + i]),  i < n/2
 * bfs_fri_commit : Fri.commit (fri.py:91-139) -- per round Merkle tree, root -> transcript (not for round 0),
 *                  alpha = xfield.sample(prover_fiat_shamir()), fold; pushes the last codeword.  Keeps the round
 *                  codewords and trees in HBM inside `session`.
 * bfs_fri_query  : sample_indices + Fri.query / Fri.query_last (fri.py:62-86, 141-176, 186-197); writes the
 *                  top-level indices (Fri.prove's return value) to h_top_level_indices[num_colinearity_tests].
 * bfs_fri_prove  : Fri.prove(codeword, proof_stream) (fri.py:178-199) = commit + query with a temporary session.
 * The codeword is limb-major in HBM (limb k at d_codeword + k*limb_stride); `ps` is a bfs_ps_new() stream, possibly
 * already holding earlier objects.  These calls synchronise `stream` (each round needs the root on the host).
 * Errors: BFS_ERR_NOT_ROOT ("omega does not have the right order", fri.py:104-105), BFS_ERR_TOO_MANY_INDICES (fri.py:69-70).
 */
int bfs_xfe_fold(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride, uint32_t log_n,
                 const uint64_t alpha[3], uint64_t offset, uint64_t omega, void* stream);
void* bfs_fri_session_new(void);
void bfs_fri_session_free(void* session);
int bfs_fri_commit(void* session, void* ps, const uint64_t* d_codeword, uint64_t limb_stride, uint32_t log_n, uint64_t offset,
                   uint64_t omega, uint32_t expansion_factor, void* stream);
int bfs_fri_query(void* session, void* ps, uint32_t num_colinearity_tests, uint64_t* h_top_level_indices, void* stream);
/* Tells the session that element `index` of round `round`'s codeword already exists in the transcript as object
 * `element_handle`: BrainfuckStark.prove pushes leaves of the combination codeword before Fri.prove opens the same
 * list (brainfuck_stark.py:325-333), and pickle writes a repeated object as a back-reference.  Call before bfs_fri_query. */
int bfs_fri_session_alias(void* session, void* ps, uint32_t round, uint64_t index, uint64_t element_handle);
/* Before bfs_fri_commit: the caller already holds the Merkle tree of the input codeword (2n digests in bfs_merkle_build_xfe's layout,
 * with its root) -- BrainfuckStark.prove commits to the combination codeword and then hands the same codeword to FRI, whose round 0
 * would hash it again (brainfuck_stark.py:301, fri.py:108).  The nodes must stay valid while the session is used. */
int bfs_fri_session_round0_tree(void* session, const uint8_t* d_nodes, const uint8_t h_root[64]);
int bfs_fri_prove(void* ps, const uint64_t* d_codeword, uint64_t limb_stride, uint32_t log_n, uint64_t offset, uint64_t omega,
                  uint32_t expansion_factor, uint32_t num_colinearity_tests, uint64_t* h_top_level_indices, void* stream);
/* wall-clock breakdown (ms) of the last commit/query on the calling thread: rounds, last codeword, Fiat-Shamir + sampling,
 * planning the openings, gather + sync, building transcript objects */
void bfs_fri_last_timing(double out[6]);
uint32_t bfs_fri_session_rounds(void* session);
int bfs_fri_session_round(void* session, uint32_t round, const uint64_t** d_codeword, uint64_t* length, uint64_t* limb_stride,
                          const uint8_t** d_nodes, uint8_t h_root[64]);

/* Device-versus-host comparison of the field primitives and compositions of them on 2^log_count operand pairs (edge values
 * first): *mismatches must come back 0; the first mismatch is described by bfs_last_error().  A test hook -- it exists
 * because a compiler fold once broke a composition of two individually correct primitives (csrc/gl.hpp, gl_sub). */
int bfs_selftest_field(uint32_t log_count, uint64_t* mismatches);

/* ---- STARK prover kernels around the transforms ------------------------------------------ */
/*
 * Codewords over the FRI domain x_i = offset * omega^i, i < n = 2^log_n, are column-major in HBM: base column c at
 * d_base + c*n, extension column c as three limb planes at d_ext + (3c + limb)*n.
 *
 * bfs_poly_randomize  Table.interpolate_columns with one randomizer (table.py:112-136): d_coeffs holds `batch`
 *     polynomials f0 of h coefficients each (the INTT of a trace column over the omicron subgroup), `stride` >= h+1
 *     apart.  Each becomes the unique interpolant of degree <= h that also takes the value h_values[b] at `point`
 *     (omega of the FRI domain): f = f0 + c (X^h - 1), c = (value - f0(point)) / (point^h - 1).  Extension columns are
 *     passed as three base polynomials with the three limbs of the random value.  Asynchronous (the values travel in the
 *     kernel arguments).
 * bfs_air_quotients   Table.all_quotients (table.py:148-168, 176-236, 249-281) of table 0..4 = processor,
 *     instruction, memory, input, output (constraints: processor_table.py:51-327, instruction_table.py:27-165,
 *     memory_table.py:45-170, io_table.py:32-75; generated into csrc/air_generated.hpp from stark_brainfuck_amd/air.py).
 *     Writes bfs_air_num_quotients(table) extension codewords (boundary, transition, terminal order) to d_out, 3n words
 *     each.  h_challenges: 11 x 3 limbs (a b c d e f alpha beta gamma delta eta), h_terminals: 5 x 3 limbs
 *     (brainfuck_stark.py:103-109), h_params: iota^(height - length) for the input/output tables (io_table.py:58-60), else NULL.
 *     height = padded table height (0 allowed: the transition quotients are then 0, table.py:181-184),
 *     unit_distance = n / height (0 when height is 0), omicron_inv = inverse of the subgroup generator.
 * bfs_difference_quotient   PermutationArgument.quotient (permutation_argument.py:9-18): (lhs - rhs) / (x - 1).
 * bfs_combination     the non-linear combination codeword (brainfuck_stark.py:236-300):
 *     out[i] = w0 * randomizer[i] + sum_s (wa_s + wb_s * x_i^shift_s) * source_s[i].  Synchronises the stream.
 */
/*
 * The Brainfuck VM and its execution trace (host): VirtualMachine.simulate (vm.py:172-306) and MemoryTable.derive_matrix
 * (memory_table.py:20-38).  program: the compiled words (vm.py:78-105), input: the symbols read by `,` as code points.
 * max_cycles: the machine stops with an error after that many cycles; 0 = BFS_VM_DEFAULT_MAX_CYCLES (a trace costs ~130 bytes per
 * cycle and a program such as `-[-]` runs for p - 1 iterations; pass UINT64_MAX to run without a limit).  Parts of the trace (bfs_vm_trace_size / bfs_vm_trace_copy, sizes in 64-bit words):
 *   0 processor matrix (rows x 7: clk ip ci ni mp mv mvi)   1 memory matrix (rows x 4: clk mp mv dummy)
 *   2 instruction matrix (rows x 3: ip ci ni, sorted by address)   3 input symbols   4 output symbols
 *   5 / 6 / 7: for the processor's memory-value column / the input symbols / the output symbols, the id of the element OBJECT
 *   that held the value in the reference (0 = the register's initial zero, 1 = the shared zero of untouched cells, k > 1 = an object
 *   made by `+`, `-`, `,`): pickle memoises by
 *   identity and these objects reach the proof through the evaluation terminals (processor_table.py:390-404).
 * Errors (BFS_ERR_BAD_ARG): unknown instruction, input exhausted ("program reads more input symbols than were supplied"),
 * cycle limit reached ("program runs for more than N cycles").
 */
#define BFS_VM_DEFAULT_MAX_CYCLES (1ull << 24)
int bfs_vm_trace_new(const uint64_t* program, size_t n_words, const uint32_t* input, size_t n_input, uint64_t max_cycles, void** trace);
void bfs_vm_trace_free(void* trace);
int bfs_vm_trace_size(void* trace, int which, size_t* words);
int bfs_vm_trace_copy(void* trace, int which, uint64_t* out);

/*
 * bfs_xfe_scan (host): the sequential column extensions of Table.extend -- processor_table.py:329-427, instruction_table.py:167-231,
 *     memory_table.py:172-206, io_table.py:77-110.  Over rows i < n with base-field columns x1, x2, x3 (NULL = absent) and an
 *     optional row mask (NULL = every row), constants c0..c3 (4 x 3 limbs):
 *       kind 0: state <- state * (c0 - c1 x1[i] - c2 x2[i] - c3 x3[i])      kind 1: state <- state * c0 + c1 x1[i] + c2 x2[i] + c3 x3[i]
 *     on masked rows; out[i], out[n + i], out[2n + i] = limbs of the state before (record_before != 0) or after row i's update;
 *     terminal = final state.
 */
int bfs_xfe_scan(int kind, const uint64_t* x1, const uint64_t* x2, const uint64_t* x3, const uint8_t* mask, uint64_t n,
                 const uint64_t constants[12], const uint64_t initial[3], int record_before, uint64_t* out, uint64_t terminal[3]);
/* host: rows x width words (row-major, rows src_stride words apart: the matrices VirtualMachine.simulate returns, vm.py:172-306) ->
 * `width` columns of `rows` words each, dst_stride words apart: the column-major form Table.pad / Table.lde work on (table.py:95-136). */
int bfs_host_transpose(const uint64_t* src, size_t rows, size_t src_stride, size_t width, uint64_t* dst, size_t dst_stride);
/*
 * bfs_xfe_scan_device: the same primitive as a prefix scan on the GPU (csrc/scan.hip): the row updates are affine maps of the
 *     running value and compose associatively.  d_x1..d_x3 / d_mask are device pointers (n words / n bytes, NULL = absent),
 *     d_x1 is read `shift1` rows ahead (cyclically), the three limb planes of the result go to d_out, d_out + out_stride,
 *     d_out + 2 out_stride.  The final state is written to d_terminal (device, three words) and / or terminal (host; this
 *     synchronises the stream); either may be NULL.
 */
int bfs_xfe_scan_device(int kind, const uint64_t* d_x1, const uint64_t* d_x2, const uint64_t* d_x3, uint64_t shift1,
                        const uint8_t* d_mask, uint64_t n, const uint64_t constants[12], const uint64_t initial[3],
                        int record_before, uint64_t* d_out, uint64_t out_stride, uint64_t* d_terminal, uint64_t* terminal, void* stream);
/* several scans in one call (a proof has nine, over short tables: three launches instead of twenty-seven).  Fields as the arguments
 * of bfs_xfe_scan_device; n >= 1.  Synchronises the stream. */
typedef struct bfs_scan_spec {
    int32_t kind, record_before;
    const uint64_t *d_x1, *d_x2, *d_x3;
    uint64_t shift1;
    const uint8_t* d_mask;
    uint64_t n;
    uint64_t constants[12];
    uint64_t initial[3];
    uint64_t* d_out;
    uint64_t out_stride;
    uint64_t* d_terminal;
} bfs_scan_spec;
int bfs_xfe_scan_device_many(const bfs_scan_spec* specs, uint32_t count, void* stream);

typedef struct bfs_comb_source {
    const uint64_t* ptr;   /* device: n words (base codeword) or 3n words (extension codeword, limb planes) */
    uint32_t is_ext, pad;
    uint64_t shift;        /* max_degree - degree bound of this codeword */
    uint64_t wa[3], wb[3]; /* weights of the plain and of the shifted term */
} bfs_comb_source;
/* Support summary of `batch` polynomials of `len` coefficients (`stride` apart): h_masks[b] has bit 63 set when the constant
 * coefficient is non-zero, and its low bits are the OR over the non-zero indices j > 0 of (j & -j) -- the lowest set bit gives the
 * power of two dividing every non-zero index, which decides which codeword elements share coefficient objects in the reference
 * (univariate.py:23-27 inside the recursive ntt).  Synchronises the stream. */
int bfs_poly_support(const uint64_t* d_coeffs, uint64_t stride, uint64_t len, uint32_t batch, uint64_t* h_masks, void* stream);
int bfs_poly_randomize(uint64_t* d_coeffs, uint64_t stride, uint64_t h, uint32_t batch, uint64_t point, const uint64_t* h_values, void* stream);
int bfs_air_num_quotients(int table);
/* the same split by kind: counts[0..2] = boundary, transition, terminal constraints of `table` (table.py:148-168, 176-236, 249-281) */
int bfs_air_counts(int table, int counts[3]);
/* The same constraints at ONE point on the host, for the verifier (brainfuck_stark.py:470-560; table.py:283-311 evaluate_*_constraints):
 * base_row / base_next: the table's base columns at the point and at the next row (next may be NULL: only the transition constraints
 * read it), ext_row / ext_next: its extension columns, 3 limbs each; out: bfs_air_num_quotients(table) x 3 limbs, boundary, transition,
 * terminal order (NOT divided by the zerofiers).  Challenges / terminals / params as for bfs_air_quotients. */
int bfs_air_evaluate(int table, const uint64_t* base_row, const uint64_t* base_next, const uint64_t* ext_row, const uint64_t* ext_next,
                     const uint64_t* h_challenges, const uint64_t* h_terminals, const uint64_t* h_params, uint64_t* out);
int bfs_air_quotients(int table, const uint64_t* d_base, const uint64_t* d_ext, uint64_t* d_out, uint32_t log_n, uint64_t unit_distance,
                      uint64_t height, uint64_t omicron_inv, uint64_t offset, uint64_t omega, const uint64_t* h_challenges,
                      const uint64_t* h_terminals, const uint64_t* h_params, void* stream);
int bfs_difference_quotient(const uint64_t* d_lhs, const uint64_t* d_rhs, uint64_t* d_out, uint32_t log_n, uint64_t offset, uint64_t omega, void* stream);
int bfs_combination(const bfs_comb_source* h_sources, uint32_t count, const uint64_t* d_randomizer, const uint64_t* h_randomizer_weight,
                    uint64_t* d_out, uint32_t log_n, uint64_t offset, uint64_t omega, void* stream);

/*
 * The same sum without materialising the quotients (the production path of BrainfuckStark.prove): nobody opens a quotient
 * codeword -- the verifier recomputes quotient values from opened trace rows (brainfuck_stark.py:470-560) -- so
 * bfs_air_combine evaluates one table's constraints at every point, divides by the zerofiers and adds
 *     sum over the table's base columns, extension columns and quotients s of (wa_s + wb_s x^shift_s) * value_s
 * to d_acc (three limb planes of n).  h_weights: base_width + ext_width + bfs_air_num_quotients(table) entries in that order.
 * d_randomizer != NULL: d_acc is initialised to w0 * randomizer first (the first call of a proof); otherwise it accumulates.
 * bfs_difference_combine adds a permutation argument's term (wa + wb x^shift) (lhs - rhs) / (x - 1).
 * Shifts must fit 32 bits.  Asynchronous on `stream` (the weights travel in the kernel arguments).
 * bfs_zerofier_inverses: every table divides by x - 1, x - omicron^-1 and x^h - 1; this computes up to 12 such denominators at every
 * point of the domain and inverts them together (one field inversion per point instead of one per table): denominator k is
 * x - h_values[k] (h_is_power[k] == 0) or x^(2^h_values[k]) - 1; d_out receives `count` codewords of n words.  bfs_air_combine takes
 * d_zerofier_inverses = {1/(x - 1), 1/(x - omicron^-1), 1/(x^height - 1)} (device codewords; the third is ignored for height 0) or
 * NULL to invert on the spot; bfs_difference_combine takes the codeword of 1/(x - 1) or NULL.
 */
typedef struct bfs_comb_weight {
    uint64_t wa[3], wb[3];
    uint64_t shift;
} bfs_comb_weight;
int bfs_air_combine(int table, const uint64_t* d_base, const uint64_t* d_ext, uint32_t log_n, uint64_t unit_distance, uint64_t height,
                    uint64_t omicron_inv, uint64_t offset, uint64_t omega, const uint64_t* h_challenges, const uint64_t* h_terminals,
                    const uint64_t* h_params, const bfs_comb_weight* h_weights, const uint64_t* d_randomizer,
                    const uint64_t* h_randomizer_weight, uint64_t* d_acc, const uint64_t* const* d_zerofier_inverses, void* stream);
int bfs_difference_combine(const uint64_t* d_lhs, const uint64_t* d_rhs, uint32_t log_n, uint64_t offset, uint64_t omega,
                           const bfs_comb_weight* h_weight, uint64_t* d_acc, const uint64_t* d_inv_x_minus_1, void* stream);
int bfs_zerofier_inverses(uint32_t log_n, uint64_t offset, uint64_t omega, uint32_t count, const uint32_t* h_is_power, const uint64_t* h_values,
                          uint64_t* d_out, void* stream);
/*
 * The same three on a RANGE of the domain's points, rows [first_row, first_row + num_rows): every rank of a cooperative proof
 * (BrainfuckStark.cooperate; brainfuck_stark.py:204-298 split by rows) holds all codewords and computes the pointwise stages for its own
 * rows only -- a row's neighbour at unit_distance is read from the rank's own copy, so there is no halo to exchange -- and the ranks then
 * all-gather the combination codeword.  d_out / d_acc are the FULL buffers (the range is written in place).
 */
int bfs_air_combine_rows(int table, const uint64_t* d_base, const uint64_t* d_ext, uint32_t log_n, uint64_t unit_distance, uint64_t height,
                         uint64_t omicron_inv, uint64_t offset, uint64_t omega, const uint64_t* h_challenges, const uint64_t* h_terminals,
                         const uint64_t* h_params, const bfs_comb_weight* h_weights, const uint64_t* d_randomizer,
                         const uint64_t* h_randomizer_weight, uint64_t* d_acc, const uint64_t* const* d_zerofier_inverses, uint64_t first_row,
                         uint64_t num_rows, void* stream);
int bfs_difference_combine_rows(const uint64_t* d_lhs, const uint64_t* d_rhs, uint32_t log_n, uint64_t offset, uint64_t omega,
                                const bfs_comb_weight* h_weight, uint64_t* d_acc, const uint64_t* d_inv_x_minus_1, uint64_t first_row,
                                uint64_t num_rows, void* stream);
int bfs_zerofier_inverses_rows(uint32_t log_n, uint64_t offset, uint64_t omega, uint32_t count, const uint32_t* h_is_power, const uint64_t* h_values,
                               uint64_t* d_out, uint64_t first_row, uint64_t num_rows, void* stream);

/* ---- BrainfuckStark.prove between its Fiat-Shamir points, as two calls (csrc/prover.cpp) ------------------------------------------
 *
 * The stages of the reference's prove() (brainfuck_stark.py:134-341) driven natively instead of call by call from the host language:
 * for a small proof the host's glue between ~110 kernel launches was two thirds of the time.
 *
 *   bfs_stark_commit   (:143-195) pads the five trace matrices (processor_table.py:24-35, instruction_table.py:19-25, memory_table.py:
 *       40-44, io_table.py:17-21), interpolates and low-degree-extends the base columns (table.py:112-148), commits to the zipped rows
 *       (:178-179), pushes the root, draws the eleven challenges (:181-183), extends the tables (the `extend` methods as prefix scans)
 *       and QUEUES the extension columns' low-degree extension; it returns while the GPU runs that.
 *       tables[5]: processor, instruction, memory, input, output matrices as VirtualMachine.simulate returns them (vm.py:172-306),
 *       row-major.  out_challenges: 11 x 3 limbs.  out_scan_terminals: 9 x 3 limbs, the final value of every extension column in the
 *       order processor (instruction permutation, memory permutation, input evaluation, output evaluation), instruction (permutation,
 *       evaluation), memory (permutation), input, output.  out_io_terminals: 2 x 3 limbs, the input / output evaluation after the last
 *       REAL row (io_table.py:106-110).  out_ms (optional, 5 doubles): host wall-clock of pad, base LDE, base tree, extension, queueing.
 *   [caller]  what hangs on object identity and symbolic degrees in the reference stays with the caller: the five terminal OBJECTS
 *       (processor_table.py:390-404; made through bfs_ps_obj_*, not pushed) and the degree bounds of all terms (:203-221, 245-293).
 *   bfs_stark_finish   (:197-336) commits to the zipped extension rows, pushes that root and the five terminal objects, draws the
 *       weights, accumulates the non-linear combination with the quotients folded in, commits to it, samples the indices, pushes the
 *       openings and runs FRI on the combination codeword.
 *       shifts: max_degree - degree bound of every term (:245-293; a quotient that vanishes has bound -1), in the reference's order
 *       (base columns, extension columns, quotients table by table, the two permutation arguments); base_field_id: the BaseField instance the base codewords' elements point at (as bfs_ps_obj_bfe);
 *       distances: the tables' distinct unit distances in the order the caller's `set` iterates them (:312); out_indices:
 *       security_level indices; out_weights_seed: 32 bytes; out_fri_indices: num_colinearity_checks; out_ms (optional, 5 doubles):
 *       extension tree, combination, its tree + indices, openings, FRI.
 * Both synchronise `stream` several times (every commitment's root goes to the host).  A session serves one proof at a time; its
 * device memory comes from the library's pool and goes back when bfs_stark_finish returns (or the session is freed).
 */
typedef struct bfs_stark_params {
    uint32_t log_n;                    /* FRI domain length 2^log_n */
    uint32_t expansion_factor, num_colinearity_checks, security_level;
    uint64_t offset, omega;            /* the FRI domain's coset offset and generator */
    uint64_t max_degree;
    uint64_t heights[3];               /* padded heights of the processor, instruction and memory tables (their constructors') */
} bfs_stark_params;
typedef struct bfs_stark_table_in {
    const uint64_t* values;            /* rows x row_stride words, row-major; the first base_width words of a row are used */
    uint64_t rows, row_stride;
} bfs_stark_table_in;
typedef struct bfs_stark_randomness {  /* every random draw of prove(), in the order it makes them (exactly one of each seed / data pair) */
    const uint8_t* randomizer_seed;    /* 32 bytes expanded on the GPU into the randomizer polynomial (bfs_xfe_sample_fill) ... */
    const uint64_t* randomizer_limbs;  /* ... or its max_degree + 1 coefficients as three limb planes (brainfuck_stark.py:162-165) */
    const uint64_t* base_randomizers;  /* one value per base column of the processor, instruction and memory tables (table.py:125-127) */
    const uint8_t* base_salt_seed;     /* 32 bytes expanded on the GPU into the salts of the base commitment (bfs_random_fill) ... */
    const uint8_t* base_salts;         /* ... or 24 bytes per leaf (salted_merkle.py:25) */
    uint64_t initials[6];              /* the two permutation arguments' initial values, 3 limbs each (brainfuck_stark.py:184-185) */
    const uint64_t* ext_randomizers;   /* three limbs per extension column of the same three tables */
    const uint8_t* ext_salt_seed;
    const uint8_t* ext_salts;
} bfs_stark_randomness;
void* bfs_stark_session_new(void);
void bfs_stark_session_free(void* session);
int bfs_stark_commit(void* session, void* ps, const bfs_stark_params* params, const bfs_stark_table_in* tables, const bfs_stark_randomness* randomness,
                     uint64_t* out_challenges, uint64_t* out_scan_terminals, uint64_t* out_io_terminals, double* out_ms, void* stream);
int bfs_stark_finish(void* session, void* ps, const uint64_t* terminal_handles, const uint64_t* terminals, const uint64_t* shifts,
                     uint32_t num_terms, int32_t base_field_id, const uint64_t* distances, uint32_t n_distances, uint64_t* out_indices,
                     uint8_t* out_weights_seed, uint64_t* out_fri_indices, double* out_ms, void* stream);

/* ---- BrainfuckStark.verify on a stream read by bfs_ps_loads (csrc/verifier.cpp; host only) -------------------------------------------
 * The reference's verifier (brainfuck_stark.py:343-579, fri.py:201-319) as two calls on the native object graph of the proof, so that a proof is
 * checked without one host-language object per pulled item:
 *   bfs_stark_verify_begin   reads the two roots and the five terminals, draws the eleven challenges (Fiat-Shamir at the reference's read
 *                            positions) and offers the prefix hashes FRI will ask for to the helper threads (params may be NULL: none offered;
 *                            only its protocol parameters are read here).  out_challenges: 11 x 3 limbs,
 *                            out_terminals: 5 x 3 limbs (canonical residues).
 *   [caller]                 the degree bounds of the 151 terms (symbolic; they depend on challenges and terminals)
 *   bfs_stark_verify_finish  weights, indices, opened rows with their salted paths, the constraints at the opened points, the inner product
 *                            against the combination leaf, FRI, the evaluation arguments against input / output / program.
 *                            shifts: max_degree - degree bound per term, in the order of bfs_stark_finish.
 * *verdict: 1 = True, 0 = False, 2 = the reference raises AssertionError (message: bfs_last_error()), 3 = the stream holds an object of a
 * kind the native checks do not model at that position: run the host-language verifier instead (it decides as the reference would).
 */
typedef struct bfs_stark_verify_params {
    uint32_t log_n, expansion_factor, num_colinearity_checks, security_level;
    uint64_t offset, omega;
    uint64_t heights[5], lengths[5], omicrons[5];      /* tables in the prover's order; lengths: unpadded (the IO tables' matter) */
    uint32_t num_distances, pad;
    uint64_t distances[8];                             /* the tables' distinct unit distances, in the caller's iteration order (:396) */
    const uint64_t* program; size_t program_len;       /* compiled words (vm.py:78-105) */
    const uint64_t* input; size_t n_input;             /* input / output symbols as code points */
    const uint64_t* output; size_t n_output;
} bfs_stark_verify_params;
int bfs_stark_verify_begin(void* ps, const bfs_stark_verify_params* params, uint64_t* out_challenges, uint64_t* out_terminals, int* verdict);
int bfs_stark_verify_finish(void* ps, const bfs_stark_verify_params* params, const uint64_t* shifts, uint32_t num_terms, int* verdict);

#ifdef __cplusplus
}
#endif
#endif /* BFSTARK_H */
