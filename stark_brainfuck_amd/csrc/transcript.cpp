// transcript.cpp -- C ABI over rp::Transcript (proof stream / Fiat-Shamir, host side).
// Declarations and reference citations: include/bfstark.h ("proof stream").
#include <cstdlib>
#include <vector>

#include "../../include/bfstark.h"
#include "blake2b.hpp"
#include "helper_pool.hpp"
#include "refpickle.hpp"
#include "runtime.hpp"

using namespace bfs;
using bfs::rp::Ref;
using bfs::rp::Transcript;

static Transcript* T(void* ps) { return (Transcript*)ps; }

static int bad_handle(uint64_t h) {
    set_error("invalid proof-stream object handle %llu", (unsigned long long)h);
    return BFS_ERR_BAD_ARG;
}

extern "C" {

void* bfs_ps_new(void) { return new Transcript(); }
void bfs_ps_free(void* ps) { delete T(ps); }      // (handing a loaded stream's teardown to a helper thread was measured: the cross-thread
                                                   //  frees made the NEXT verification 2-3 x slower -- tools/verify_time.py, profiles/r05/README.md)

// The result is only handed out if serialising it gives the input back byte for byte; otherwise (an opcode or an object kind outside
// what the reference's proofs contain, or a pickle some other writer laid out differently) NULL with bfs_last_error() saying why --
// the caller falls back to CPython's unpickler + bfs_ps_obj_*.
void* bfs_ps_loads(const uint8_t* data, size_t len) {
    std::string why;
    rp::Unpickler u(data, len);
    Ref root = u.load(&why);
    if (!root) { set_error("bfs_ps_loads: %s", why.c_str()); return nullptr; }
    if (root->kind != rp::K_LIST) { set_error("bfs_ps_loads: the pickle is not a list"); return nullptr; }
    Transcript* t = new Transcript();
    for (const Ref& item : root->items) {
        t->add(item);                      // handle = index + 1, as bfs_ps_object_at reports it
        t->objects.push_back(item);
    }
    t->loaded_from_bytes = true;
    std::string again;
    try {
        again = t->serialize(t->objects.size());
    } catch (const std::exception& e) {       // (the writer refuses a graph it cannot walk: never seen from the reader, kept as a net)
        delete t;
        set_error("bfs_ps_loads: %s", e.what());
        return nullptr;
    }
    if (again.size() != len || memcmp(again.data(), data, len) != 0) {
        delete t;
        set_error("bfs_ps_loads: the stream does not serialise back to the same %zu bytes", len);
        return nullptr;
    }
    return t;
}

uint64_t bfs_ps_obj_bytes(void* ps, const uint8_t* data, size_t len) { return T(ps)->add(rp::mk_bytes(data, len)); }
uint64_t bfs_ps_obj_int(void* ps, uint64_t value) { return T(ps)->add(rp::mk_int(value)); }
uint64_t bfs_ps_obj_xfe(void* ps, const uint64_t limbs[3]) {
    uint64_t l[3] = {limbs[0] % GL_P, limbs[1] % GL_P, limbs[2] % GL_P};
    return T(ps)->add(T(ps)->world.xfe_compact(l));
}
uint64_t bfs_ps_obj_bfe(void* ps, uint64_t value, int field_id) {
    if (field_id < 0 || field_id > 64) { set_error("BaseField instance id %d out of range", field_id); return 0; }
    return T(ps)->add(T(ps)->world.bfe_in(value % GL_P, T(ps)->world.base_field(field_id)));
}
uint64_t bfs_ps_obj_xfe_from(void* ps, const uint64_t* coefficient_handles, size_t n) {
    if (n > 3) { set_error("an extension element has at most 3 coefficients"); return 0; }
    std::vector<Ref> coeffs;
    for (size_t i = 0; i < n; ++i) {
        Ref r = T(ps)->get(coefficient_handles[i]);
        if (!r || r->role != rp::R_BFE) { bad_handle(coefficient_handles[i]); return 0; }
        coeffs.push_back(r);
    }
    if (n && coeffs[n - 1]->limbs[0] == 0) { set_error("leading coefficient of an extension element must be non-zero (extension_field.py:6-9)"); return 0; }
    return T(ps)->add(T(ps)->world.xfe_from(coeffs));
}

static uint64_t make_seq(void* ps, rp::Kind kind, const uint64_t* handles, size_t n) {
    std::vector<Ref> items;
    items.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        Ref r = T(ps)->get(handles[i]);
        if (!r) { bad_handle(handles[i]); return 0; }
        items.push_back(r);
    }
    Ref node = rp::mk(kind);
    node->items = items;
    return T(ps)->add(node);
}
uint64_t bfs_ps_obj_list(void* ps, const uint64_t* handles, size_t n) { return make_seq(ps, rp::K_LIST, handles, n); }
uint64_t bfs_ps_obj_tuple(void* ps, const uint64_t* handles, size_t n) { return make_seq(ps, rp::K_TUPLE, handles, n); }

int bfs_ps_push(void* ps, uint64_t handle) {
    Ref r = T(ps)->get(handle);
    if (!r) return bad_handle(handle);
    T(ps)->objects.push_back(r);
    return BFS_OK;
}
size_t bfs_ps_num_objects(void* ps) { return T(ps)->objects.size(); }

uint64_t bfs_ps_object_at(void* ps, size_t index) {
    Transcript* t = T(ps);
    if (index >= t->objects.size()) { set_error("proof stream has %zu objects, asked for %zu", t->objects.size(), index); return 0; }
    const rp::Node* want = t->objects[index].get();
    // pushed objects always come from the arena; search from the back (recent objects are the common case)
    for (size_t i = t->arena.size(); i-- > 0;)
        if (t->arena[i].get() == want) return i + 1;
    return t->add(t->objects[index]);
}

int bfs_ps_serialize(void* ps, size_t count, uint8_t* out, size_t capacity, size_t* length) {
    std::string s = T(ps)->serialize(count);
    *length = s.size();
    if (out && capacity >= s.size()) memcpy(out, s.data(), s.size());
    return BFS_OK;
}

int bfs_ps_obj_dumps(void* ps, uint64_t handle, uint8_t* out, size_t capacity, size_t* length) {
    Ref r = T(ps)->get(handle);
    if (!r) return bad_handle(handle);
    rp::Pickler p(&T(ps)->world);
    std::string s = p.dumps(r);
    *length = s.size();
    if (out && capacity >= s.size()) memcpy(out, s.data(), s.size());
    return BFS_OK;
}

// Merkle.verify / SaltedMerkle.verify (merkle.py:54-63, salted_merkle.py:55-68) on objects of this stream: the leaf is
// blake2b(pickle.dumps(element) [+ pickle.dumps(salt)]), every path node is hashed to the left or right of the running digest by the
// parity of the index, and the result must equal `root`.  One call per opening instead of two pickles and depth + 1 hashes through the
// host language (a proof has ~600 of them).  *ok: 1 accepted, 0 rejected.  The objects may be anything the stream holds: a path that is
// not a list of byte strings is simply rejected, as the reference's `running + node` would raise.
int bfs_ps_merkle_verify(void* ps, uint64_t element_handle, uint64_t salt_handle, uint64_t path_handle, uint64_t index, const uint8_t* root,
                         size_t root_len, int* ok) {
    *ok = 0;
    Ref element = T(ps)->get(element_handle), path = T(ps)->get(path_handle);
    if (!element) return bad_handle(element_handle);
    if (!path) return bad_handle(path_handle);
    rp::Pickler p(&T(ps)->world);
    std::string pre = p.dumps(element);
    if (salt_handle) {
        Ref salt = T(ps)->get(salt_handle);
        if (!salt) return bad_handle(salt_handle);
        pre += p.dumps(salt);
    }
    if (path->kind != rp::K_LIST) return BFS_OK;
    unsigned char running[64];
    blake2b_host(pre.data(), pre.size(), running);
    std::string buf;
    for (const Ref& node : path->items) {
        if (!node || node->kind != rp::K_BYTES) return BFS_OK;
        buf.clear();
        if ((index & 1) == 0) { buf.append((const char*)running, 64); buf.append(node->bytes(), node->nbytes()); }
        else { buf.append(node->bytes(), node->nbytes()); buf.append((const char*)running, 64); }
        blake2b_host(buf.data(), buf.size(), running);
        index >>= 1;
    }
    *ok = (root_len == 64 && memcmp(running, root, 64) == 0) ? 1 : 0;
    return BFS_OK;
}

// sum_i weights[i] * terms[i] over the cubic extension (3 limbs each, canonical or not: reduced on the way in): the verifier's
// inner product of brainfuck_stark.py:553-554 (303 products per opened index)
int bfs_xfe_inner_product(const uint64_t* weights, const uint64_t* terms, size_t count, uint64_t out[3]) {
    Xfe acc{{0, 0, 0}};
    for (size_t i = 0; i < count; ++i) {
        const Xfe w{{weights[3 * i] % GL_P, weights[3 * i + 1] % GL_P, weights[3 * i + 2] % GL_P}};
        const Xfe t{{terms[3 * i] % GL_P, terms[3 * i + 1] % GL_P, terms[3 * i + 2] % GL_P}};
        acc = xfe_add(acc, xfe_mul(w, t));
    }
    out[0] = acc.c[0]; out[1] = acc.c[1]; out[2] = acc.c[2];
    return BFS_OK;
}

// the verifier knows (roughly) at which read positions it will ask: the hashes over those prefixes are offered to the helper threads
// (only for a stream made by bfs_ps_loads, which no longer changes).  Returns the number of hashes offered; 0 is not an error.
size_t bfs_ps_prefetch_fiat_shamir(void* ps, const size_t* counts, size_t n, size_t num_bytes) {
    return T(ps)->prefetch_fiat_shamir(counts, n, num_bytes);
}

int bfs_ps_fiat_shamir(void* ps, size_t count, uint8_t* out, size_t num_bytes) {
    T(ps)->fiat_shamir(count, out, num_bytes);
    return BFS_OK;
}

// push a 64-byte digest and return the Fiat-Shamir bytes over the stream including it, the way the FRI prover does while a tree
// kernel runs: placeholder pushed and everything in front of its payload absorbed first, the digest filled in afterwards
int bfs_ps_push_digest_fiat_shamir(void* ps, const uint8_t digest[64], uint8_t* out, size_t num_bytes) {
    rp::Transcript::Speculation sp;
    T(ps)->speculate(sp);
    T(ps)->resolve(sp, digest, out, num_bytes);
    return BFS_OK;
}

// the same for a run of digests, the way bfs_fri_commit does it when the stream is long: the pickles of all `count` coming streams are
// made first and helper threads absorb their prefixes (Transcript::Lookahead); out: count * num_bytes bytes
int bfs_ps_push_digests_fiat_shamir(void* ps, const uint8_t* digests, size_t count, uint8_t* out, size_t num_bytes, int* used_lookahead) {
    rp::Transcript::Lookahead look;
    const bool on = T(ps)->lookahead_begin(look, count, 0);
    if (used_lookahead) *used_lookahead = on ? 1 : 0;
    for (size_t k = 0; k < count; ++k) {
        if (on) T(ps)->lookahead_next(look, digests + 64 * k, out + num_bytes * k, num_bytes);
        else bfs_ps_push_digest_fiat_shamir(ps, digests + 64 * k, out + num_bytes * k, num_bytes);
    }
    return BFS_OK;
}

int bfs_ps_obj_kind(void* ps, uint64_t handle) {
    Ref r = T(ps)->get(handle);
    if (!r) return -1;
    if (r->kind == rp::K_XFE) return 100;
    if (r->kind == rp::K_INSTANCE) return r->role == rp::R_XFE ? 100 : (r->role == rp::R_BFE ? 101 : 102);
    return (int)r->kind;
}
size_t bfs_ps_obj_len(void* ps, uint64_t handle) {
    Ref r = T(ps)->get(handle);
    if (!r) return 0;
    return (r->kind == rp::K_BYTES || r->kind == rp::K_STR) ? r->nbytes() : r->items.size();
}
uint64_t bfs_ps_obj_item(void* ps, uint64_t handle, size_t i) {
    Transcript* t = T(ps);
    Ref r = t->get(handle);
    if (!r || i >= r->items.size()) return 0;
    const rp::Node* want = r->items[i].get();
    for (size_t k = t->arena.size(); k-- > 0;)
        if (t->arena[k].get() == want) return k + 1;
    return t->add(r->items[i]);
}
int bfs_ps_obj_get_bytes(void* ps, uint64_t handle, uint8_t* out, size_t capacity) {
    Ref r = T(ps)->get(handle);
    if (!r || capacity < r->nbytes()) return bad_handle(handle);
    memcpy(out, r->bytes(), r->nbytes());
    return BFS_OK;
}
int bfs_ps_obj_get_limbs(void* ps, uint64_t handle, uint64_t limbs[3]) {
    Ref r = T(ps)->get(handle);
    if (!r) return bad_handle(handle);
    if (r->kind == rp::K_INT) { limbs[0] = r->ival; limbs[1] = limbs[2] = 0; return BFS_OK; }
    for (int i = 0; i < 3; ++i) limbs[i] = r->limbs[i];
    return BFS_OK;
}

/* BrainfuckStark.sample_weights (brainfuck_stark.py:104-112): weight i = ExtensionField.sample(blake2b(randomness + bytes(i)).digest()),
 * bytes(i) being i zero bytes; out: 3 * count limbs */
int bfs_sample_weights(const uint8_t* randomness, size_t len, size_t count, uint64_t* out) {
    // message i = randomness followed by i zero bytes: all messages share their full 128-byte blocks with the longer ones, so the
    // chaining value after k full blocks is computed once (a proof draws ~300 weights from messages of up to three blocks: one
    // compression per weight instead of up to three)
    std::vector<unsigned char> msg(len + count + 128, 0);
    if (len) memcpy(msg.data(), randomness, len);
    u64 chain[8];                                           // state after `full` blocks of the common prefix
    blake2b_init(chain);
    size_t full = 0;
    for (size_t i = 0; i < count; ++i) {
        const size_t L = len + i;
        // blocks before the last one: a message of L bytes has ceil(L / 128) blocks (one, empty, for L = 0); the last is final
        const size_t before = L ? (L - 1) / 128 : 0;
        while (full < before) {
            u64 m[16];
            memcpy(m, msg.data() + 128 * full, 128);
            ++full;
            blake2b_compress(chain, m, 128 * full, false);
        }
        u64 h[8], m[16];
        memcpy(h, chain, sizeof h);
        memset(m, 0, sizeof m);
        if (L - 128 * before) memcpy(m, msg.data() + 128 * before, L - 128 * before);
        blake2b_compress(h, m, L, true);
        const Xfe x = rp::sample_xfe((const unsigned char*)h, 64);          // three chunks of 64 // 3 = 21 bytes; the 64th byte is not used
        for (int k = 0; k < 3; ++k) out[3 * i + k] = x.c[k];
    }
    return BFS_OK;
}

/* BaseField.sample / ExtensionField.sample (algebra.py:138-142, extension_field.py:100-111) */
uint64_t bfs_gl_sample(const uint8_t* bytes, size_t len) { return rp::sample_base(bytes, len); }
void bfs_xfe_sample(const uint8_t* bytes, size_t len, uint64_t out[3]) {
    Xfe x = rp::sample_xfe(bytes, len);
    for (int i = 0; i < 3; ++i) out[i] = x.c[i];
}

}  // extern "C"
