// verifier.cpp -- BrainfuckStark.verify on a proof stream read natively (bfs_stark_verify_begin / bfs_stark_verify_finish), host only.
//
// The reference's verifier (/root/reference/code/brainfuck_stark.py:343-579 with fri.py:201-319) pulls ~1 200 objects off the stream, checks
// ~60 authentication paths, recomputes the 303-term non-linear combination at the opened points and runs FRI's colinearity checks.
// stark_brainfuck_amd/brainfuck_stark.py (_verify_stream) and fri.py (Fri.verify) mirror it in Python; this file is the same sequence of checks on
// the node graph bfs_ps_loads read from the proof bytes, so that a proof is checked without one Python object per pulled item.  Two calls, like the
// prover's, because the degree bounds of the 151 terms are the caller's (symbolic, multivariate.py:144-170) and depend on what the first call reads:
//   bfs_stark_verify_begin    the two commitments' roots and the five terminals -> challenges (Fiat-Shamir at the reference's read positions)
//   bfs_stark_verify_finish   weights, indices, opened rows + salted paths, constraints at the opened points (bfs_air_evaluate), inner product
//                             against the combination leaf, FRI, evaluation arguments against the public input / output / program
// Verdict 1 / 0 = the reference's True / False; 2 = the reference raises an AssertionError (message: bfs_last_error); 3 = the stream holds
// something this file does not model (an object of an unexpected kind where the reference would raise some other exception, coinciding
// abscissae in a colinearity check, ...): the caller then runs the Python verifier, which decides as the reference would.  The native path
// therefore only ever answers for streams whose every pulled object has the expected shape.
#include "../../include/bfstark.h"

#include "blake2b.hpp"
#include "refpickle.hpp"
#include "runtime.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace bfs;
using bfs::rp::Ref;
using bfs::rp::Transcript;

namespace {

enum { V_FALSE = 0, V_TRUE = 1, V_ASSERT = 2, V_FALLBACK = 3 };

struct Fallback {};                                        // thrown by the readers below: an object is not what its position requires
struct Assertion { std::string message; };
struct Reject {};                                          // the reference returns False

struct Reader {
    Transcript* t;
    size_t at = 0;
    Ref pull() {                                            // ip.py:12-16
        if (at >= t->objects.size()) throw Assertion{"ProofStream: cannot pull object; queue empty."};
        return t->objects[at++];
    }
    void fiat_shamir(unsigned char out[32]) { t->fiat_shamir(at, out, 32); }      // ip.py:27-30: over objects[:read_index]
};

const Ref& bytes_of(const Ref& r) {
    if (!r || r->kind != rp::K_BYTES) throw Fallback{};
    return r;
}
// limbs of an element as canonical residues (what the reference's arithmetic sees: every operation reduces, algebra.py:89-99)
Xfe value_of(const Ref& r) {
    if (!r || r->kind != rp::K_INSTANCE) throw Fallback{};
    if (r->role == rp::R_XFE) return Xfe{{r->limbs[0] % GL_P, r->limbs[1] % GL_P, r->limbs[2] % GL_P}};
    if (r->role == rp::R_BFE) return Xfe{{r->limbs[0] % GL_P, 0, 0}};
    throw Fallback{};
}
Xfe xfe_value_of(const Ref& r) {
    if (!r || r->kind != rp::K_INSTANCE || r->role != rp::R_XFE) throw Fallback{};
    return Xfe{{r->limbs[0] % GL_P, r->limbs[1] % GL_P, r->limbs[2] % GL_P}};
}
bool xfe_is_zero(const Xfe& a) { return !(a.c[0] | a.c[1] | a.c[2]); }
bool xfe_eq(const Xfe& a, const Xfe& b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2]; }

// Merkle.verify / SaltedMerkle.verify (merkle.py:54-63, salted_merkle.py:55-68)
bool path_ok(Transcript* t, const Ref& element, const Ref* salt, const Ref& path, u64 index, const Ref& root) {
    if (!path || path->kind != rp::K_LIST) throw Fallback{};
    rp::Pickler p(&t->world);
    std::string pre = p.dumps(element);
    if (salt) pre += p.dumps(*salt);
    unsigned char running[64];
    blake2b_host(pre.data(), pre.size(), running);
    std::string buf;
    for (const Ref& node : path->items) {
        if (!node || node->kind != rp::K_BYTES) throw Fallback{};
        buf.clear();
        if ((index & 1) == 0) { buf.append((const char*)running, 64); buf.append(node->bytes(), node->nbytes()); }
        else { buf.append(node->bytes(), node->nbytes()); buf.append((const char*)running, 64); }
        blake2b_host(buf.data(), buf.size(), running);
        index >>= 1;
    }
    return root->nbytes() == 64 && memcmp(running, root->bytes(), 64) == 0;
}

// int.from_bytes(blake2b(seed + bytes(counter)).digest(), "big") % size
u64 sample_index(std::vector<unsigned char>& msg, u64 size) {
    unsigned char digest[64];
    blake2b_host(msg.data(), msg.size(), digest);
    msg.push_back(0);
    u128 acc = 0;
    for (int i = 0; i < 64; ++i) acc = ((acc << 8) | digest[i]) % size;
    return (u64)acc;
}

Xfe xfe_pow_u(Xfe a, u64 e) {
    Xfe acc{{1, 0, 0}};
    while (e) {
        if (e & 1) acc = xfe_mul(acc, a);
        a = xfe_mul(a, a);
        e >>= 1;
    }
    return acc;
}

struct Begin {
    Ref base_root, ext_root;
    Xfe terminals[5];          // reduced: what the reference's arithmetic sees
    Xfe terminals_stored[5];   // as stored: what its comparisons see (Polynomial.__eq__ / BaseFieldElement.__eq__ compare .value, algebra.py:48-49)
    u64 challenges[33];
};

void read_begin(Reader& rd, Begin& b) {
    b.base_root = bytes_of(rd.pull());
    unsigned char seed[32];
    rd.fiat_shamir(seed);
    if (bfs_sample_weights(seed, 32, 11, b.challenges) != BFS_OK) throw Fallback{};
    b.ext_root = bytes_of(rd.pull());
    for (int k = 0; k < 5; ++k) {
        const Ref t = rd.pull();
        b.terminals[k] = xfe_value_of(t);
        b.terminals_stored[k] = Xfe{{t->limbs[0], t->limbs[1], t->limbs[2]}};
    }
}

// fri.py:201-319 (stark_brainfuck_amd/fri.py: Fri.verify), on the reader's stream
bool fri_verify(Reader& rd, const bfs_stark_verify_params& P, const Ref& root0) {
    const u64 N = 1ull << P.log_n;
    u32 rounds = 0;
    for (u64 len = N; len > P.expansion_factor; len /= 2) ++rounds;
    const u32 t = P.num_colinearity_checks;
    std::vector<Ref> roots{root0};
    std::vector<Xfe> alphas;
    for (u32 r = 0; r < rounds; ++r) {
        if (r > 0) roots.push_back(bytes_of(rd.pull()));
        unsigned char seed[32];
        rd.fiat_shamir(seed);
        alphas.push_back(rp::sample_xfe(seed, 32));
    }
    Ref last = rd.pull();
    if (!last || last->kind != rp::K_LIST) throw Fallback{};
    const size_t n_last = last->items.size();
    if (n_last == 0 || (n_last & (n_last - 1))) throw Fallback{};          // (never a FRI codeword; the reference's padding rules live in Merkle)
    std::vector<Xfe> last_values;
    {
        // the last codeword hashes to the last root (fri.py:236-241)
        rp::Pickler p(&rd.t->world);
        std::vector<std::string> level;
        for (const Ref& e : last->items) {
            last_values.push_back(xfe_value_of(e));
            const std::string pre = p.dumps(e);
            unsigned char d[64];
            blake2b_host(pre.data(), pre.size(), d);
            level.emplace_back((const char*)d, 64);
        }
        while (level.size() > 1) {
            std::vector<std::string> next;
            for (size_t i = 0; i + 1 < level.size(); i += 2) {
                const std::string both = level[i] + level[i + 1];
                unsigned char d[64];
                blake2b_host(both.data(), both.size(), d);
                next.emplace_back((const char*)d, 64);
            }
            level.swap(next);
        }
        const Ref& want = roots.back();
        if (want->nbytes() != 64 || memcmp(want->bytes(), level[0].data(), 64) != 0) return false;
    }
    // its interpolant has degree <= n_last / expansion - 1 (fri.py:243-259).  Coefficient j of the interpolant is, up to a non-zero factor,
    // sum_i y_i omega^(-i j)
    u64 omega = P.omega % GL_P, offset = P.offset % GL_P;
    u64 last_omega = omega;
    for (u32 r = 0; r + 1 < rounds; ++r) last_omega = gl_mul(last_omega, last_omega);
    if (gl_pow(last_omega, n_last) != 1) throw Assertion{"omega does not have right order"};
    {
        const long degree = (long)(n_last / P.expansion_factor) - 1;
        const u64 inverse = gl_inv(last_omega);
        long top = -1;
        for (long j = (long)n_last - 1; j >= 0 && top < 0; --j) {
            const u64 step = gl_pow(inverse, (u64)j);
            u64 w = 1;
            Xfe acc{{0, 0, 0}};
            for (const Xfe& y : last_values) {
                acc = xfe_add(acc, xfe_scale(y, w));
                w = gl_mul(w, step);
            }
            if (!xfe_is_zero(acc)) top = j;
        }
        if (top > degree) return false;
    }
    // indices (fri.py:62-86)
    std::vector<u64> top_level;
    {
        const u64 size = N >> 1, reduced_size = N >> (rounds - 1);
        if (t > reduced_size) {
            char msg[160];
            snprintf(msg, sizeof msg, "cannot sample more indices than available in last codeword; requested: %u, available: %llu", t, (unsigned long long)reduced_size);
            throw Assertion{msg};
        }
        unsigned char seed[32];
        rd.fiat_shamir(seed);
        std::vector<unsigned char> msg(seed, seed + 32);
        std::vector<u64> reduced;
        while (top_level.size() < t) {
            const u64 index = sample_index(msg, size), red = index % reduced_size;
            bool seen = false;
            for (u64 x : reduced) seen |= x == red;
            if (!seen) { top_level.push_back(index); reduced.push_back(red); }
        }
    }
    for (u32 r = 0; r + 1 < rounds; ++r) {
        const u64 half = N >> (r + 1);
        std::vector<u64> c_idx(t), a_idx(t), b_idx(t);
        for (u32 s = 0; s < t; ++s) { c_idx[s] = top_level[s] % half; a_idx[s] = c_idx[s]; b_idx[s] = c_idx[s] + half; }
        const Xfe alpha = alphas[r];
        std::vector<Ref> aa(t), bb(t), cc(t);
        for (u32 s = 0; s < t; ++s) {
            Ref triple = rd.pull();
            if (!triple || triple->kind != rp::K_TUPLE || triple->items.size() != 3) throw Fallback{};
            aa[s] = triple->items[0]; bb[s] = triple->items[1]; cc[s] = triple->items[2];
            const u64 ax = gl_mul(offset, gl_pow(omega, a_idx[s])), bx = gl_mul(offset, gl_pow(omega, b_idx[s]));
            const Xfe ya = xfe_value_of(aa[s]), yb = xfe_value_of(bb[s]), yc = xfe_value_of(cc[s]);
            // three points on a line: (ax, ya), (bx, yb), (alpha, yc)  (univariate.py: test_colinearity; stark_brainfuck_amd/fri.py: _on_a_line)
            const u64 d1 = gl_sub(bx, ax);
            const Xfe d2 = xfe_sub_base(alpha, ax), dx = xfe_sub_base(alpha, bx);
            if (!d1 || xfe_is_zero(d2) || xfe_is_zero(dx)) throw Fallback{};       // coinciding abscissae: the general routine decides
            const Xfe e1 = xfe_sub(yb, ya);
            const bool on_a_line = !xfe_is_zero(e1) && xfe_eq(xfe_scale(xfe_sub(yc, ya), d1), xfe_mul(e1, d2));
            if (!on_a_line) return false;
        }
        for (u32 i = 0; i < t; ++i) {
            if (!path_ok(rd.t, aa[i], nullptr, rd.pull(), a_idx[i], roots[r])) return false;
            if (!path_ok(rd.t, bb[i], nullptr, rd.pull(), b_idx[i], roots[r])) return false;
            if (r + 1 != rounds - 1 && !path_ok(rd.t, cc[i], nullptr, rd.pull(), c_idx[i], roots[r + 1])) return false;
        }
        if (r + 1 == rounds - 1) {
            // fri.py:311-315: the folded values are elements of the last codeword (object comparison: coefficient values as stored)
            for (u32 i = 0; i < t; ++i) {
                const Ref& e = last->items[c_idx[i]];
                if (cc[i]->role != rp::R_XFE || e->role != rp::R_XFE) throw Fallback{};
                if (cc[i]->limbs[0] != e->limbs[0] || cc[i]->limbs[1] != e->limbs[1] || cc[i]->limbs[2] != e->limbs[2]) return false;
            }
        }
        omega = gl_mul(omega, omega);
        offset = gl_mul(offset, offset);
    }
    return true;
}

constexpr int NT = 5;
constexpr u32 BASE_W[NT] = {7, 3, 4, 1, 1};
constexpr u32 EXT_W[NT] = {4, 2, 1, 1, 1};

int verify_finish(Transcript* t, const bfs_stark_verify_params& P, const u64* shifts, u32 num_terms) {
    Reader rd{t};
    Begin b;
    read_begin(rd, b);
    const u64 n = 1ull << P.log_n;
    u32 num_base = 0, num_ext = 0, num_quot = 0;
    int nb[NT], nt[NT], nz[NT];
    for (int k = 0; k < NT; ++k) {
        num_base += BASE_W[k]; num_ext += EXT_W[k];
        int counts[3];
        if (bfs_air_counts(k, counts) != BFS_OK) throw Fallback{};
        nb[k] = counts[0]; nt[k] = counts[1]; nz[k] = counts[2];
        num_quot += (u32)(nb[k] + nt[k] + nz[k]);
    }
    num_quot += 2;
    if (num_terms != num_base + num_ext + num_quot) throw Fallback{};
    const u32 num_weights = 1 + 2 * num_terms;
    unsigned char seed[32];
    rd.fiat_shamir(seed);
    std::vector<u64> weights(3ull * num_weights);
    if (bfs_sample_weights(seed, 32, num_weights, weights.data()) != BFS_OK) throw Fallback{};
    const Ref comb_root = bytes_of(rd.pull());
    rd.fiat_shamir(seed);
    std::vector<u64> indices;
    {
        std::vector<unsigned char> msg(seed, seed + 32);
        for (u32 i = 0; i < P.security_level; ++i) indices.push_back(sample_index(msg, n));
    }
    // opened rows (brainfuck_stark.py:383-412)
    std::map<u64, std::vector<Xfe>> rows;
    for (u64 index : indices)
        for (u32 d = 0; d <= P.num_distances; ++d) {
            const u64 idx = (index + (d ? P.distances[d - 1] : 0)) % n;
            std::vector<Xfe> row;
            for (int which = 0; which < 2; ++which) {
                Ref element = rd.pull();
                Ref sp = rd.pull();
                if (!element || element->kind != rp::K_TUPLE || !sp || sp->kind != rp::K_TUPLE || sp->items.size() != 2) throw Fallback{};
                if (element->items.size() != (which == 0 ? 1 + num_base : num_ext)) throw Fallback{};
                if (!path_ok(t, element, &sp->items[0], sp->items[1], idx, which == 0 ? b.base_root : b.ext_root))
                    throw Assertion{which == 0 ? "salted base tree verify must succeed for base codewords"
                                               : "salted base tree verify must succeed for extension codewords"};
                for (size_t k = 0; k < element->items.size(); ++k) {
                    const Ref& e = element->items[k];
                    if (which == 1 || k == 0) row.push_back(xfe_value_of(e)); else row.push_back(value_of(e));
                }
            }
            rows[idx] = row;
        }
    u64 terminals_flat[15];
    for (int k = 0; k < 5; ++k) for (int l = 0; l < 3; ++l) terminals_flat[3 * k + l] = b.terminals[k].c[l];
    const u64 offset = P.offset % GL_P, omega = P.omega % GL_P;
    for (u64 index : indices) {
        const u64 x = gl_mul(offset, gl_pow(omega, index));
        auto shifted = [&](const Xfe& v, u32 term) { return xfe_scale(v, gl_pow(x, shifts[term])); };
        const std::vector<Xfe>& row = rows[index];
        std::vector<Xfe> terms;
        terms.reserve(num_weights);
        terms.push_back(row[0]);
        u32 term = 0;
        for (u32 i = 0; i < num_base + num_ext; ++i, ++term) { terms.push_back(row[1 + i]); terms.push_back(shifted(row[1 + i], term)); }
        const u64 boundary_inverse = gl_inv(gl_sub(x, 1));
        u32 bcol = 1, ecol = 1 + num_base;
        Xfe proc_ext[2], instr_ext0{{0, 0, 0}}, mem_ext0{{0, 0, 0}};
        for (int k = 0; k < NT; ++k) {
            const u64 h = P.heights[k];
            const u64 unit = h ? n / h : 0;
            const std::vector<Xfe>& nrow = rows[(index + unit) % n];
            if (nrow.size() != row.size()) throw Fallback{};
            u64 br[8], bn[8], er[12], en[12], out[3 * 32], params[3];
            for (u32 c = 0; c < BASE_W[k]; ++c) { br[c] = row[bcol + c].c[0]; bn[c] = nrow[bcol + c].c[0]; }
            for (u32 c = 0; c < EXT_W[k]; ++c)
                for (int l = 0; l < 3; ++l) { er[3 * c + l] = row[ecol + c].c[l]; en[3 * c + l] = nrow[ecol + c].c[l]; }
            if (k == 0) { proc_ext[0] = row[ecol]; proc_ext[1] = row[ecol + 1]; }
            if (k == 1) instr_ext0 = row[ecol];
            if (k == 2) mem_ext0 = row[ecol];
            const u64* pr = nullptr;
            if (k >= 3) {                                   // io_table.py:58-60: iota^(height - length)
                const u64* iota = b.challenges + 3 * (k == 3 ? 8 : 9);
                const Xfe v = xfe_pow_u(Xfe{{iota[0], iota[1], iota[2]}}, h - P.lengths[k]);
                params[0] = v.c[0]; params[1] = v.c[1]; params[2] = v.c[2];
                pr = params;
            }
            if (bfs_air_evaluate(k, br, bn, er, en, b.challenges, terminals_flat, pr, out) != BFS_OK) throw Fallback{};
            const u64 omicron_inverse = gl_inv(P.omicrons[k] % GL_P);
            const u64 transition_factor = h ? gl_mul(gl_sub(x, omicron_inverse), gl_inv(gl_sub(gl_pow(x, h), 1))) : 0;
            const u64 terminal_inverse = gl_inv(gl_sub(x, omicron_inverse));
            for (int q = 0; q < nb[k] + nt[k] + nz[k]; ++q, ++term) {
                const u64 factor = q < nb[k] ? boundary_inverse : (q < nb[k] + nt[k] ? transition_factor : terminal_inverse);
                const Xfe v = xfe_scale(Xfe{{out[3 * q], out[3 * q + 1], out[3 * q + 2]}}, factor);
                terms.push_back(v);
                terms.push_back(shifted(v, term));
            }
            bcol += BASE_W[k]; ecol += EXT_W[k];
        }
        {   // permutation arguments (brainfuck_stark.py:62-65; permutation_argument.py:20-34)
            const Xfe q0 = xfe_scale(xfe_sub(proc_ext[0], instr_ext0), boundary_inverse);
            terms.push_back(q0); terms.push_back(shifted(q0, term)); ++term;
            const Xfe q1 = xfe_scale(xfe_sub(proc_ext[1], mem_ext0), boundary_inverse);
            terms.push_back(q1); terms.push_back(shifted(q1, term)); ++term;
        }
        if (terms.size() != num_weights) throw Fallback{};
        Xfe inner{{0, 0, 0}};
        for (u32 i = 0; i < num_weights; ++i)
            inner = xfe_add(inner, xfe_mul(Xfe{{weights[3 * i], weights[3 * i + 1], weights[3 * i + 2]}}, terms[i]));
        Ref leaf = rd.pull();
        Ref path = rd.pull();
        if (!path_ok(t, leaf, nullptr, path, index, comb_root)) return V_FALSE;
        // brainfuck_stark.py:567 compares the leaf OBJECT with the inner product: coefficient values as stored
        if (!leaf || leaf->kind != rp::K_INSTANCE || leaf->role != rp::R_XFE) return V_FALSE;
        if (leaf->limbs[0] != inner.c[0] || leaf->limbs[1] != inner.c[1] || leaf->limbs[2] != inner.c[2]) return V_FALSE;
    }
    bool verdict = fri_verify(rd, P, comb_root);
    // evaluation arguments (evaluation_argument.py:1-53): the input / output symbols in gamma / delta, the program rows in eta
    auto C = [&](int i) { return Xfe{{b.challenges[3 * i], b.challenges[3 * i + 1], b.challenges[3 * i + 2]}}; };
    auto horner_symbols = [&](const u64* s, size_t count, const Xfe& point) {
        Xfe acc{{0, 0, 0}};
        for (size_t i = 0; i < count; ++i) acc = xfe_add_base(xfe_mul(acc, point), s[i] % GL_P);
        return acc;
    };
    // `ea.select_terminal(terminals) == ea.compute_terminal(challenges)` (brainfuck_stark.py:574-577) compares the terminal OBJECT of the
    // proof with a computed (canonical) element, coefficient values as stored: a terminal whose limb is written as v + p is unequal in the
    // reference although every other use of it reduces -- so the three comparisons below take the stored limbs (round-5 advice)
    verdict = verdict && xfe_eq(b.terminals_stored[2], horner_symbols(P.input, P.n_input, C(8)));
    verdict = verdict && xfe_eq(b.terminals_stored[3], horner_symbols(P.output, P.n_output, C(9)));
    {
        const Xfe a = C(0), bb = C(1), c = C(2), eta = C(10);
        Xfe acc{{0, 0, 0}};
        for (size_t address = 0; address <= P.program_len; ++address) {
            Xfe row = xfe_scale(a, (u64)address % GL_P);
            if (address < P.program_len) {
                const u64 word = P.program[address] % GL_P, next = address + 1 < P.program_len ? P.program[address + 1] % GL_P : 0;
                row = xfe_add(xfe_add(row, xfe_scale(bb, word)), xfe_scale(c, next));
            }
            acc = xfe_add(xfe_mul(acc, eta), row);
        }
        verdict = verdict && xfe_eq(b.terminals_stored[4], acc);
    }
    return verdict ? V_TRUE : V_FALSE;
}

template <class F>
int guarded(F&& body, int* verdict) {
    try {
        *verdict = body();
    } catch (const Fallback&) {
        *verdict = V_FALLBACK;
    } catch (const Reject&) {
        *verdict = V_FALSE;
    } catch (const Assertion& a) {
        set_error("%s", a.message.c_str());
        *verdict = V_ASSERT;
    } catch (const std::exception& e) {
        set_error("bfs_stark_verify: %s", e.what());
        return BFS_ERR_BAD_ARG;
    }
    return BFS_OK;
}

}  // namespace

extern "C" {

int bfs_stark_verify_begin(void* ps, const bfs_stark_verify_params* params, uint64_t* out_challenges, uint64_t* out_terminals, int* verdict) {
    Transcript* t = (Transcript*)ps;
    if (!t || !t->loaded_from_bytes) { set_error("bfs_stark_verify_begin: the stream was not read by bfs_ps_loads"); return BFS_ERR_BAD_ARG; }
    return guarded([&]() {
        Reader rd{t};
        Begin b;
        read_begin(rd, b);
        // Fiat-Shamir ahead of time.  The read positions at which FRI asks are known from the protocol parameters alone: behind the
        // openings (four objects per opened row, two per combination leaf) it asks before every round's root and once behind the last
        // codeword -- each time over ~50 KB.  Exactly those prefixes go to the helper threads now, in the order they will be needed; the
        // three early positions (1, 7, 8 objects) are a block or two each and are hashed when asked for.
        if (params != nullptr) {
            u32 rounds = 0;
            for (u64 len = 1ull << params->log_n; len > params->expansion_factor; len /= 2) ++rounds;
            const size_t p0 = 8 + (size_t)params->security_level * (1 + params->num_distances) * 4 + (size_t)params->security_level * 2;
            std::vector<size_t> at;
            for (u32 r = 0; r <= rounds; ++r)
                if (p0 + r <= t->objects.size()) at.push_back(p0 + r);
            if (!at.empty()) (void)t->prefetch_fiat_shamir(at.data(), at.size(), 32);
        }
        memcpy(out_challenges, b.challenges, sizeof b.challenges);
        for (int k = 0; k < 5; ++k) for (int l = 0; l < 3; ++l) out_terminals[3 * k + l] = b.terminals[k].c[l];
        return (int)V_TRUE;
    }, verdict);
}

int bfs_stark_verify_finish(void* ps, const bfs_stark_verify_params* params, const uint64_t* shifts, uint32_t num_terms, int* verdict) {
    Transcript* t = (Transcript*)ps;
    if (!t || !t->loaded_from_bytes) { set_error("bfs_stark_verify_finish: the stream was not read by bfs_ps_loads"); return BFS_ERR_BAD_ARG; }
    if (params->log_n < 2 || params->log_n > 32 || params->num_distances > 8 || params->expansion_factor < 2) { set_error("bfs_stark_verify_finish: parameters"); return BFS_ERR_BAD_ARG; }
    for (uint32_t s = 0; s < num_terms; ++s)
        if (shifts[s] >> 32) { set_error("bfs_stark_verify_finish: shift of term %u does not fit 32 bits", s); return BFS_ERR_BAD_ARG; }
    return guarded([&]() { return verify_finish(t, *params, shifts, num_terms); }, verdict);
}

}  // extern "C"
