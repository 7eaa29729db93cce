// refpickle.hpp -- host-side transcript of the reference's proof stream, byte for byte.
//
// The reference's ProofStream is a Python list; its serialisation is pickle.dumps(list) and the Fiat-Shamir
// challenge is shake_256 of those bytes (/root/reference/code/ip.py:18-25).  The bytes depend on CPython's
// pickler: memoisation by object identity, opcode choice, batching and 64 KiB framing.  This file re-creates
// that for the object kinds the hot path pushes (bytes, ints, lists, tuples, field elements) by
//   (1) building the same object GRAPH the reference would hold -- classes, interned attribute-name strings,
//       the shared BaseField / ExtensionField instances, one node per Python object, identity = node address;
//   (2) walking it with a small pickler that follows Modules/_pickle.c (protocol 4): save(), memo_get/put,
//       save_long, save_bytes, save_unicode, save_tuple, batch_list_exact, batch_dict_exact, save_global,
//       save_reduce(copyreg.__newobj__), _Pickler_OpcodeBoundary / _Pickler_CommitFrame.
// Pinned by tests against pickle byte strings captured from the reference (tests/golden/pickle.json, fri_*_stream.bin).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <functional>
#include <memory>
#include <stdexcept>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "gl.hpp"
#include "helper_pool.hpp"
#include "keccak.hpp"

namespace bfs {
namespace rp {

enum Kind { K_BYTES = 0, K_INT = 1, K_STR = 2, K_LIST = 3, K_TUPLE = 4, K_DICT = 5, K_CLASS = 6, K_INSTANCE = 7,
            K_XFE = 8 /* ExtensionFieldElement kept as three limbs; the pickler expands it (one allocation instead of ~12) */ };
enum Role { R_NONE = 0, R_XFE = 1, R_BFE = 2 };  // what an INSTANCE node stands for (for reading values back)

struct Node;
typedef std::shared_ptr<Node> Ref;

struct Node {
    Kind kind = K_INT;
    Role role = R_NONE;
    std::string data;        // BYTES / STR payload
    u64 ival = 0;            // INT (0 <= v < 2^64)
    std::vector<Ref> items;  // LIST / TUPLE elements; DICT: k0 v0 k1 v1 ...; CLASS: module, qualname
    Ref cls, state;          // INSTANCE
    u64 limbs[3] = {0, 0, 0};  // R_XFE: value; R_BFE: limbs[0]
    unsigned char* inl = nullptr;   // BYTES of exactly 64 bytes (digests: most nodes of a proof) live behind the node itself, see DigestNode
    const char* bytes() const { return inl ? (const char*)inl : data.data(); }
    size_t nbytes() const { return inl ? 64 : data.size(); }
    void set_bytes(const void* p, size_t len) {
        if (inl && len == 64) { memcpy(inl, p, 64); return; }
        inl = nullptr;
        data.assign((const char*)p, len);
    }
    unsigned depth = 1;      // Unpickler only: 1 + the deepest child at the time the children were attached (early refusal of deep nesting)
    // Teardown without recursion: a node that dies hands its children to a worklist owned by the outermost destructor on this thread,
    // so a chain of a million nested tuples (which the Unpickler can be made to build from two bytes per level) is freed in a loop
    // instead of a million nested ~Node calls (round-4 advice: a 2 MB proof killed the verifier's process with a stack overflow).
    ~Node() {
        if (items.empty() && !cls && !state) return;
        static thread_local std::vector<Ref>* pending = nullptr;
        if (pending) { release_children(*pending); return; }
        std::vector<Ref> work;
        pending = &work;
        release_children(work);
        while (!work.empty()) {
            Ref r = std::move(work.back());
            work.pop_back();
            r.reset();               // a last reference: ~Node of that child runs with `pending` set and only appends to `work`
        }
        pending = nullptr;
    }
    Node() = default;
    Node(const Node&) = delete;
    Node& operator=(const Node&) = delete;

   private:
    void release_children(std::vector<Ref>& out) {
        for (Ref& r : items) if (r) out.push_back(std::move(r));
        items.clear();
        if (cls) out.push_back(std::move(cls));
        if (state) out.push_back(std::move(state));
    }
};
// a 64-byte BYTES node in ONE allocation (node + payload) instead of two (node, std::string buffer): a FRI proof makes ~2 500 of them
struct DigestNode : Node {
    unsigned char payload[64];
};

inline Ref mk(Kind k) { Ref n = std::make_shared<Node>(); n->kind = k; return n; }
inline Ref mk_int(u64 v) { Ref n = mk(K_INT); n->ival = v; return n; }
inline Ref mk_str(const char* s) { Ref n = mk(K_STR); n->data = s; return n; }
inline Ref mk_bytes(const void* p, size_t len) {
    if (len == 64) {
        std::shared_ptr<DigestNode> d = std::make_shared<DigestNode>();
        d->kind = K_BYTES;
        d->inl = d->payload;
        memcpy(d->payload, p, 64);
        return d;
    }
    Ref n = mk(K_BYTES);
    n->data.assign((const char*)p, len);
    return n;
}
inline Ref mk_list(const std::vector<Ref>& it) { Ref n = mk(K_LIST); n->items = it; return n; }
inline Ref mk_list(std::vector<Ref>&& it) { Ref n = mk(K_LIST); n->items = std::move(it); return n; }
inline Ref mk_tuple(const std::vector<Ref>& it) { Ref n = mk(K_TUPLE); n->items = it; return n; }
inline Ref mk_class(const Ref& module, const Ref& name) { Ref n = mk(K_CLASS); n->items = {module, name}; return n; }
inline Ref mk_instance(const Ref& cls, const std::vector<Ref>& kv) {
    Ref n = mk(K_INSTANCE);
    n->cls = cls;
    n->state = mk(K_DICT);
    n->state->items = kv;
    return n;
}

// The shared part of the reference's object graph: what `ExtensionField.main()` creates once
// (extension_field.py:88-98) plus the classes and the interned attribute names.
struct World {
    Ref s_value, s_field, s_p, s_coefficients, s_polynomial, s_modulus;
    Ref c_bfe, c_bf, c_poly, c_xfe, c_xf;
    Ref bf_internal;    // the BaseField instance inside the ExtensionField's modulus ("variant A")
    Ref bf_standalone;  // a BaseField.main() of its own (pure base-field leaves)
    Ref xfield;

    World() {
        Ref m_alg = mk_str("algebra"), m_uni = mk_str("univariate"), m_ext = mk_str("extension_field");
        s_value = mk_str("value"); s_field = mk_str("field"); s_p = mk_str("p");
        s_coefficients = mk_str("coefficients"); s_polynomial = mk_str("polynomial"); s_modulus = mk_str("modulus");
        c_bfe = mk_class(m_alg, mk_str("BaseFieldElement"));
        c_bf = mk_class(m_alg, mk_str("BaseField"));
        c_poly = mk_class(m_uni, mk_str("Polynomial"));
        c_xfe = mk_class(m_ext, mk_str("ExtensionFieldElement"));
        c_xf = mk_class(m_ext, mk_str("ExtensionField"));
        bf_internal = mk_instance(c_bf, {s_p, mk_int(GL_P)});
        bf_standalone = mk_instance(c_bf, {s_p, mk_int(GL_P)});
        Ref one = bfe(1, true), minus_one = bfe(GL_P - 1, true), zero = bfe(0, true);
        Ref modulus = mk_instance(c_poly, {s_coefficients, mk_list({one, minus_one, zero, one})});  // same `one` twice
        xfield = mk_instance(c_xf, {s_modulus, modulus});
    }

    // BaseFieldElement(value, field)  algebra.py:15-18
    Ref bfe(u64 v, bool internal) const { return bfe_in(v, internal ? bf_internal : bf_standalone); }
    Ref bfe_in(u64 v, const Ref& field) const {
        Ref n = mk_instance(c_bfe, {s_value, mk_int(v), s_field, field});
        n->role = R_BFE;
        n->limbs[0] = v;
        return n;
    }

    // further BaseField.main() instances: the reference creates one per class that asks for it (vm.py:70, brainfuck_stark.py:21)
    // and elements keep pointing at "their" instance, which pickle writes out separately.  id 0 = bf_standalone,
    // 1 = bf_internal, >= 2 created on first use.
    std::vector<Ref> bf_extra;
    Ref base_field(int id) {
        if (id == 0) return bf_standalone;
        if (id == 1) return bf_internal;
        while ((int)bf_extra.size() <= id - 2) bf_extra.push_back(mk_instance(c_bf, {s_p, mk_int(GL_P)}));
        return bf_extra[id - 2];
    }

    // ExtensionFieldElement over explicit coefficient objects (shared BaseFieldElement objects, foreign BaseField instances)
    Ref xfe_from(const std::vector<Ref>& coeffs) const {
        Ref poly = mk_instance(c_poly, {s_coefficients, mk_list(coeffs)});
        Ref n = mk_instance(c_xfe, {s_polynomial, poly, s_field, xfield});
        n->role = R_XFE;
        for (size_t i = 0; i < 3; ++i) n->limbs[i] = i < coeffs.size() ? coeffs[i]->limbs[0] : 0;
        return n;
    }

    // compact form of xfe(): same pickle bytes, built lazily by Pickler::save_xfe
    Ref xfe_compact(const u64 limbs[3]) const {
        Ref n = mk(K_XFE);
        n->role = R_XFE;
        for (int i = 0; i < 3; ++i) n->limbs[i] = limbs[i];
        return n;
    }

    // ExtensionFieldElement(Polynomial(coeffs trimmed of trailing zeros), xfield)  extension_field.py:5-9
    Ref xfe(const u64 limbs[3]) const {
        int k = limbs[2] ? 3 : (limbs[1] ? 2 : (limbs[0] ? 1 : 0));
        std::vector<Ref> coeffs;
        for (int i = 0; i < k; ++i) coeffs.push_back(bfe(limbs[i], true));
        Ref poly = mk_instance(c_poly, {s_coefficients, mk_list(coeffs)});
        Ref n = mk_instance(c_xfe, {s_polynomial, poly, s_field, xfield});
        n->role = R_XFE;
        for (int i = 0; i < 3; ++i) n->limbs[i] = limbs[i];
        return n;
    }
};

class Pickler {
   public:
    explicit Pickler(const World* world = nullptr) : world_(world) {}
    std::string dumps(const Ref& root) {
        out_.clear();
        memo_.clear();
        memo_next_ = 0;
        int_marks.clear();
        frame_start_ = NPOS;
        framing_ = false;
        const unsigned char proto[2] = {0x80, 4};
        write(proto, 2);
        framing_ = true;
        save(root.get());
        op(0x2e);  // STOP
        commit_frame();
        framing_ = false;
        return out_;
    }

    // pickle.dumps(items[:count]) without building that list: the verifier's prefix hashes run on helper threads, and copying a thousand
    // shared pointers per prefix there meant a thousand contended reference-count updates on nodes every other thread is walking too
    std::string dumps_prefix(const std::vector<Ref>& items, size_t count) {
        if (count > items.size()) count = items.size();
        Node list_object;                                   // stands for the list itself: the first memo slot
        list_object.kind = K_LIST;
        out_.clear();
        memo_.clear();
        memo_next_ = 0;
        int_marks.clear();
        frame_start_ = NPOS;
        framing_ = false;
        const unsigned char proto[2] = {0x80, 4};
        write(proto, 2);
        framing_ = true;
        op(0x5d);  // EMPTY_LIST
        memo_put(&list_object);
        save_list_items(items.data(), count);
        op(0x2e);  // STOP
        commit_frame();
        framing_ = false;
        memo_.clear();                                      // (holds the address of a stack object)
        return out_;
    }

    // ---- incremental pickling of a growing list (the proof stream): pickle.dumps(objects) for objects[:k] is a prefix-stable
    // byte string -- memo indices are handed out in order, so the encoding of the first k objects never changes -- followed by
    // APPENDS STOP and the frame length patched in.  Valid from two objects on (a one-element list has no MARK, _pickle.c
    // batch_list_exact).  stream_begin / stream_item build the open form, stream_bytes returns the closed pickle of what has
    // been added so far and leaves the open form intact.
    void stream_begin() {
        out_.clear();
        memo_.clear();
        memo_next_ = 0;
        int_marks.clear();
        frame_start_ = NPOS;
        framing_ = false;
        const unsigned char proto[2] = {0x80, 4};
        write(proto, 2);
        framing_ = true;
        op(0x5d);                      // EMPTY_LIST
        memo_put(&stream_list_);
        stream_items_ = 0;
        stream_batch_ = 0;
    }
    void stream_item(const Ref& item) {
        if (stream_batch_ == 0) op(0x28);                      // MARK opens a batch
        save(item.get());
        ++stream_items_;
        if (++stream_batch_ == BATCH) { op(0x65); stream_batch_ = 0; }   // APPENDS closes it after 1000 items
    }
    size_t stream_items() const { return stream_items_; }
    size_t stream_size() const { return out_.size(); }
    // the open form can be extended tentatively: stream_mark(), stream_item() of objects that are not part of the stream (yet),
    // stream_bytes(), then stream_rollback(mark, those objects) -- Transcript::Lookahead pickles the NEXT rounds' streams that way
    struct StreamMark { size_t size, frame, items, batch; uint32_t memo_next; std::string header; };
    StreamMark stream_mark() const {
        StreamMark m{out_.size(), frame_start_, stream_items_, stream_batch_, memo_next_, std::string()};
        if (frame_start_ != NPOS) m.header = out_.substr(frame_start_, FRAME_HEADER);
        return m;
    }
    void stream_rollback(const StreamMark& m, const std::vector<Ref>& tentative) {
        for (const Ref& r : tentative) memo_.erase(r.get());
        memo_next_ = m.memo_next;
        out_.resize(m.size);
        if (m.frame != NPOS) out_.replace(m.frame, FRAME_HEADER, m.header);
        frame_start_ = m.frame;
        stream_items_ = m.items;
        stream_batch_ = m.batch;
    }
    // overwrite bytes of the open form (same offsets as in the last stream_bytes() result): a payload that was pickled as a
    // placeholder and has become known (Transcript::speculate / resolve)
    void stream_patch(size_t offset, const void* data, size_t len) { memcpy(&out_[offset], data, len); }
    std::string stream_bytes() {
        std::string result;
        stream_bytes_into(result);
        return result;
    }
    void stream_bytes_into(std::string& result) {      // (a caller that keeps `result` around saves the allocation and its page faults)
        stream_closed([&](const std::string& closed) { result.assign(closed); });
    }
    // bytes [from, end) of what stream_bytes() returns, and the nine bytes at `watch` in it (the header of a frame that is still open
    // there depends on everything behind it)
    void stream_tail_into(std::string& tail, size_t from, size_t watch, unsigned char watched[9]) {
        stream_closed([&](const std::string& closed) {
            tail.assign(closed, from < closed.size() ? from : closed.size(), std::string::npos);
            for (size_t i = 0; i < FRAME_HEADER; ++i) watched[i] = watch + i < closed.size() ? (unsigned char)closed[watch + i] : 0;
        });
    }
    // where the header of the frame lies that the next item goes into
    size_t stream_open_frame() const { return frame_start_ != NPOS ? frame_start_ : out_.size(); }

   private:
    // run `use` on the finished pickle of the items so far, then reopen the stream for more items
    template <class F>
    void stream_closed(F&& use) {
        const size_t size = out_.size(), frame = frame_start_;
        std::string saved_header;
        if (frame != NPOS) saved_header = out_.substr(frame, FRAME_HEADER);
        if (stream_batch_ != 0) op(0x65);                      // APPENDS of the open batch
        op(0x2e);                                              // STOP
        commit_frame();                                        // (a frame is open in any case after the two writes)
        use(out_);
        out_.resize(size);                                     // (frame == NPOS: the closing opcodes opened a frame of their own, all of it goes)
        if (frame != NPOS) out_.replace(frame, FRAME_HEADER, saved_header);   // reopen the frame the items live in
        frame_start_ = frame;
    }

   private:
    const World* world_;
    Node stream_list_;
    size_t stream_items_ = 0, stream_batch_ = 0;
    static constexpr size_t NPOS = (size_t)-1;
    static constexpr size_t FRAME_HEADER = 9, FRAME_MIN = 4, FRAME_TARGET = 64 * 1024, BATCH = 1000;
    std::string out_;
    // pickle's memo: object -> index.  Open addressing on the node address (a proof pickles ~10^4 nodes, and the node-per-entry
    // std::unordered_map was most of the pickling time); inner objects of a compact element only advance the index counter.
    struct PtrMap {
        std::vector<const Node*> keys;
        std::vector<uint32_t> vals;
        size_t used = 0;                       // occupied slots including tombstones
        static const Node* tomb() { return reinterpret_cast<const Node*>(uintptr_t(1)); }
        void clear() { keys.clear(); vals.clear(); used = 0; }
        size_t slot_of(const Node* k) const { return (size_t)(((uintptr_t)k * 0x9E3779B97F4A7C15ULL) >> 17) & (keys.size() - 1); }
        bool find(const Node* k, uint32_t& v) const {
            if (keys.empty()) return false;
            for (size_t j = slot_of(k);; j = (j + 1) & (keys.size() - 1)) {
                if (keys[j] == k) { v = vals[j]; return true; }
                if (keys[j] == nullptr) return false;
            }
        }
        void grow() {
            std::vector<const Node*> k2(keys.empty() ? 1024 : 2 * keys.size(), nullptr);
            std::vector<uint32_t> v2(k2.size(), 0);
            keys.swap(k2); vals.swap(v2);
            used = 0;
            for (size_t i = 0; i < k2.size(); ++i)
                if (k2[i] != nullptr && k2[i] != tomb()) put(k2[i], v2[i]);
        }
        void put(const Node* k, uint32_t v) {          // k is not in the map
            if (2 * (used + 1) > keys.size()) grow();
            size_t j = slot_of(k);
            while (keys[j] != nullptr && keys[j] != tomb()) j = (j + 1) & (keys.size() - 1);
            if (keys[j] == nullptr) ++used;
            keys[j] = k; vals[j] = v;
        }
        void erase(const Node* k) {
            if (keys.empty()) return;
            for (size_t j = slot_of(k); keys[j] != nullptr; j = (j + 1) & (keys.size() - 1))
                if (keys[j] == k) { keys[j] = tomb(); return; }
        }
    };
    PtrMap memo_;
    uint32_t memo_next_ = 0;
    size_t frame_start_ = NPOS;
    bool framing_ = false;

    void write(const void* p, size_t n) {  // _Pickler_Write
        if (framing_ && frame_start_ == NPOS) {
            frame_start_ = out_.size();
            out_.append(FRAME_HEADER, (char)0xFE);
        }
        out_.append((const char*)p, n);
    }
    void op(unsigned char c) { write(&c, 1); }
    void commit_frame() {  // _Pickler_CommitFrame
        if (!framing_ || frame_start_ == NPOS) return;
        size_t len = out_.size() - frame_start_ - FRAME_HEADER;
        if (len >= FRAME_MIN) {
            out_[frame_start_] = (char)0x95;
            uint64_t l = len;
            memcpy(&out_[frame_start_ + 1], &l, 8);
        } else {
            out_.erase(frame_start_, FRAME_HEADER);
        }
        frame_start_ = NPOS;
    }
    void opcode_boundary() {  // _Pickler_OpcodeBoundary
        if (!framing_ || frame_start_ == NPOS) return;
        if (out_.size() - frame_start_ - FRAME_HEADER >= FRAME_TARGET) commit_frame();
    }
    void memo_put(const Node* n) {
        memo_.put(n, memo_next_++);
        op(0x94);  // MEMOIZE
    }
    void memo_skip() {   // an inner object of a compact element: takes a memo slot, can never be referenced again
        ++memo_next_;
        op(0x94);
    }

    // byte-for-byte what save() emits for World::xfe(limbs): instance -> dict{polynomial: instance(dict{coefficients:
    // [BaseFieldElement...]}), field: xfield}; inner objects only consume memo indices
    void save_xfe(const Node* n) {
        const World& w = *world_;
        const int k = n->limbs[2] ? 3 : (n->limbs[1] ? 2 : (n->limbs[0] ? 1 : 0));
        save(w.c_xfe.get()); op(0x29); op(0x81); memo_put(n);
        opcode_boundary(); op(0x7d); memo_skip();                       // state dict of the element
        op(0x28);
        save(w.s_polynomial.get());
        opcode_boundary(); save(w.c_poly.get()); op(0x29); op(0x81); memo_skip();   // Polynomial instance
        opcode_boundary(); op(0x7d); memo_skip();                       // its dict (one item -> SETITEM)
        save(w.s_coefficients.get());
        opcode_boundary(); op(0x5d); memo_skip();                       // coefficient list
        if (k > 1) op(0x28);
        for (int i = 0; i < k; ++i) {
            opcode_boundary(); save(w.c_bfe.get()); op(0x29); op(0x81); memo_skip();   // BaseFieldElement
            opcode_boundary(); op(0x7d); memo_skip();
            op(0x28);
            save(w.s_value.get());
            opcode_boundary(); save_long(n->limbs[i]);
            save(w.s_field.get());
            save(w.bf_internal.get());
            op(0x75); op(0x62);                                         // SETITEMS, BUILD
            if (k == 1) op(0x61);                                       // APPEND (single item)
        }
        if (k > 1) op(0x65);                                            // APPENDS
        op(0x73); op(0x62);                                             // SETITEM (coefficients), BUILD (Polynomial)
        save(w.s_field.get());
        save(w.xfield.get());
        op(0x75); op(0x62);                                             // SETITEMS, BUILD (element)
    }

    void memo_get(uint32_t idx) {
        if (idx < 256) { unsigned char b[2] = {0x68, (unsigned char)idx}; write(b, 2); }
        else { unsigned char b[5] = {0x6a}; memcpy(b + 1, &idx, 4); write(b, 5); }
    }

   public:
    // (offset, length, value) of every integer opcode written by the last dumps(): used to cut a pickle into a template
    // around its variable-length integers (rows.hip)
    struct IntMark { size_t offset, length; u64 value; };
    std::vector<IntMark> int_marks;
    bool record_ints = false;

   private:
    void save_long(u64 v) {
        const size_t mark_at = out_.size() + ((framing_ && frame_start_ == NPOS) ? FRAME_HEADER : 0);
        save_long_raw(v);
        if (record_ints) int_marks.push_back(IntMark{mark_at, out_.size() - mark_at, v});
    }
    void save_long_raw(u64 v) {
        unsigned char b[12];
        if (v < (1ull << 31)) {
            uint32_t x = (uint32_t)v;
            if (x >> 16) { b[0] = 0x4a; memcpy(b + 1, &x, 4); write(b, 5); }
            else if (x >> 8) { b[0] = 0x4d; b[1] = (unsigned char)x; b[2] = (unsigned char)(x >> 8); write(b, 3); }
            else { b[0] = 0x4b; b[1] = (unsigned char)x; write(b, 2); }
            return;
        }
        int bits = 64 - __builtin_clzll(v);
        int nn = bits / 8 + 1;
        b[0] = 0x8a; b[1] = (unsigned char)nn;
        memset(b + 2, 0, 10);
        memcpy(b + 2, &v, 8);
        write(b, 2 + (size_t)nn);
    }

    // the items of a list as _pickle.c's batch_list_exact writes them: one item -> APPEND, more -> MARK ... APPENDS in batches of 1000
    void save_list_items(const Ref* items, size_t len) {
        if (len == 1) { save(items[0].get()); op(0x61); }  // APPEND
        else if (len > 1) {
            size_t total = 0;
            do {
                size_t batch = 0;
                op(0x28);  // MARK
                while (total < len) {
                    save(items[total].get());
                    ++total;
                    if (++batch == BATCH) break;
                }
                op(0x65);  // APPENDS
            } while (total < len);
        }
    }
    void save(const Node* n) {
        if (n == nullptr) throw std::runtime_error("a null object in the graph");        // (never from a graph the reader or the C ABI built)
        opcode_boundary();
        if (n->kind == K_INT) { save_long(n->ival); return; }
        uint32_t seen;
        if (memo_.find(n, seen)) { memo_get(seen); return; }
        switch (n->kind) {
            case K_XFE: save_xfe(n); break;
            case K_BYTES: {
                size_t len = n->nbytes();
                if (len < 256) { unsigned char h[2] = {0x43, (unsigned char)len}; write(h, 2); }
                else { unsigned char h[5] = {0x42}; uint32_t l = (uint32_t)len; memcpy(h + 1, &l, 4); write(h, 5); }
                write(n->bytes(), len);
                memo_put(n);
                break;
            }
            case K_STR: {
                size_t len = n->data.size();
                if (len < 256) { unsigned char h[2] = {0x8c, (unsigned char)len}; write(h, 2); }
                else { unsigned char h[5] = {0x58}; uint32_t l = (uint32_t)len; memcpy(h + 1, &l, 4); write(h, 5); }
                write(n->data.data(), len);
                memo_put(n);
                break;
            }
            case K_LIST: {
                op(0x5d);  // EMPTY_LIST
                memo_put(n);
                save_list_items(n->items.data(), n->items.size());
                break;
            }
            case K_TUPLE: {
                size_t len = n->items.size();
                if (len == 0) { op(0x29); break; }  // EMPTY_TUPLE, not memoised
                if (len <= 3) {
                    for (auto& x : n->items) save(x.get());
                    op((unsigned char)(0x84 + len));  // TUPLE1/2/3
                } else {
                    op(0x28);
                    for (auto& x : n->items) save(x.get());
                    op(0x74);  // TUPLE
                }
                memo_put(n);
                break;
            }
            case K_DICT: {
                op(0x7d);  // EMPTY_DICT
                memo_put(n);
                size_t pairs = n->items.size() / 2;
                if (pairs == 1) { save(n->items[0].get()); save(n->items[1].get()); op(0x73); }  // SETITEM
                else if (pairs > 1) {
                    size_t i = 0;
                    while (i < pairs) {
                        size_t batch = 0;
                        op(0x28);
                        while (i < pairs) {
                            save(n->items[2 * i].get());
                            save(n->items[2 * i + 1].get());
                            ++i;
                            if (++batch == BATCH) break;
                        }
                        op(0x75);  // SETITEMS
                    }
                }
                break;
            }
            case K_CLASS: {  // save_global, protocol 4
                save(n->items[0].get());
                save(n->items[1].get());
                op(0x93);  // STACK_GLOBAL
                memo_put(n);
                break;
            }
            case K_INSTANCE: {  // save_reduce with copyreg.__newobj__(cls), state = __dict__
                save(n->cls.get());
                op(0x29);  // args[1:] == ()
                op(0x81);  // NEWOBJ
                memo_put(n);
                save(n->state.get());
                op(0x62);  // BUILD
                break;
            }
            default: break;
        }
    }
};

// The other direction: a protocol-4 pickle of the kinds above -> the node graph (ProofStream.deserialize, ip.py:27-30 on the
// verifier's side).  One node per unpickled object, memo references (BINGET / LONG_BINGET) become shared nodes, so the graph has the
// identities the writer's objects had and the Pickler above writes the same bytes back -- which is how bfs_ps_loads checks the result
// before anybody uses it.  Opcodes outside this set (None, booleans, negative integers, floats, reducers other than
// copyreg.__newobj__ with no arguments) make load() fail, and the caller takes the Python route.
class Unpickler {
   public:
    Unpickler(const unsigned char* data, size_t len) : p_(data), n_(len) {}
    // the unpickled object, or an empty Ref (`why` says what was not understood)
    Ref load(std::string* why = nullptr) {
        Ref r = run();
        if (r) {
            const char* bad = check_shape(r);
            if (bad) { r.reset(); why_ = bad; }
        }
        if (!r) {
            // a refused graph may hold reference cycles (a list appended to itself through the memo): every cycle passes through a
            // memoised node, so emptying those lets the shared pointers free the rest (iteratively, ~Node)
            for (Ref& m : memo_) { m->items.clear(); m->cls.reset(); m->state.reset(); }
            stack_.clear();
            memo_.clear();
            if (why) *why = why_;
        }
        return r;
    }

    // The Pickler walks the graph recursively, and the bytes come from whoever wrote the proof: a stream nested a million lists deep
    // must be refused here, not overflow the stack there; and a graph with a cycle (which pickle can express and the reference's
    // proofs never contain) would never be freed by reference counting.  Iterative depth-first walk along the pickler's own traversal
    // (children in order; a finished node is a memo reference and is not entered again).  nullptr when the graph is fine.
    static const char* check_shape(const Ref& root, size_t limit = 200) {
        struct Frame { const Node* n; size_t next; };
        std::vector<Frame> path;
        std::unordered_map<const Node*, char> state;                 // 1: on the current path, 2: finished
        auto child = [](const Node* n, size_t i) -> const Node* {
            if (i < n->items.size()) return n->items[i].get();
            i -= n->items.size();
            if (i == 0) return n->state.get();
            return n->cls.get();
        };
        if (!root) return nullptr;
        path.push_back({root.get(), 0});
        state[root.get()] = 1;
        while (!path.empty()) {
            Frame& f = path.back();
            if (f.next >= f.n->items.size() + 2) { state[f.n] = 2; path.pop_back(); continue; }
            const Node* c = child(f.n, f.next++);
            if (c == nullptr || c->kind == K_INT) continue;
            auto it = state.find(c);
            if (it != state.end()) {
                if (it->second == 1) return "the object graph has a cycle";
                continue;
            }
            if (path.size() + 1 > limit) return "objects nested deeper than 200 levels";
            state[c] = 1;
            path.push_back({c, 0});
        }
        return nullptr;
    }

   private:
    const unsigned char* p_;
    size_t n_, pos_ = 0;
    std::vector<Ref> stack_, memo_;
    std::vector<Ref> instances_;       // every object NEWOBJ made: each must have received its state by STOP (see there)
    std::vector<size_t> marks_;
    std::string why_;

    bool need(size_t k) {
        if (n_ - pos_ < k) { why_ = "truncated pickle"; return false; }
        return true;
    }
    u64 le(size_t k) {
        u64 v = 0;
        for (size_t i = 0; i < k; ++i) v |= (u64)p_[pos_ + i] << (8 * i);
        pos_ += k;
        return v;
    }
    bool pop(Ref& r) {
        if (stack_.empty() || (!marks_.empty() && stack_.size() <= marks_.back())) { why_ = "stack underflow"; return false; }
        r = stack_.back();
        stack_.pop_back();
        return true;
    }
    bool pop_mark(std::vector<Ref>& items) {
        if (marks_.empty()) { why_ = "no MARK"; return false; }
        const size_t m = marks_.back();
        marks_.pop_back();
        items.assign(stack_.begin() + m, stack_.end());
        stack_.resize(m);
        return true;
    }
    bool blob(size_t len_bytes, Kind kind) {
        if (!need(len_bytes)) return false;
        const u64 len = le(len_bytes);
        if (!need(len)) return false;
        Ref n;
        if (kind == K_BYTES) n = mk_bytes(p_ + pos_, len);
        else { n = mk(K_STR); n->data.assign((const char*)p_ + pos_, len); }
        pos_ += len;
        stack_.push_back(n);
        return true;
    }
    static bool is_class(const Ref& c, const char* module, const char* name) {
        return c && c->kind == K_CLASS && c->items.size() == 2 && c->items[0]->data == module && c->items[1]->data == name;
    }
    static Ref dict_get(const Ref& d, const char* key) {
        if (!d || d->kind != K_DICT) return Ref();
        for (size_t i = 0; i + 1 < d->items.size(); i += 2)
            if (d->items[i]->kind == K_STR && d->items[i]->data == key) return d->items[i + 1];
        return Ref();
    }
    // what an instance stands for, for bfs_ps_obj_kind / bfs_ps_obj_get_limbs (the pickler does not look at these)
    static void tag(const Ref& inst) {
        if (is_class(inst->cls, "algebra", "BaseFieldElement")) {
            Ref v = dict_get(inst->state, "value");
            if (v && v->kind == K_INT) { inst->role = R_BFE; inst->limbs[0] = v->ival; }
        } else if (is_class(inst->cls, "extension_field", "ExtensionFieldElement")) {
            Ref poly = dict_get(inst->state, "polynomial");
            Ref coeffs = poly && poly->kind == K_INSTANCE ? dict_get(poly->state, "coefficients") : Ref();
            if (coeffs && coeffs->kind == K_LIST && coeffs->items.size() <= 3) {
                bool all_base = true;                       // every coefficient a BaseFieldElement with an integer value (else: no role --
                for (const Ref& c : coeffs->items)          // readers of limbs then see "not an element" instead of a silent zero)
                    all_base = all_base && c && c->kind == K_INSTANCE && c->role == R_BFE;
                if (all_base) {
                    inst->role = R_XFE;
                    for (size_t i = 0; i < coeffs->items.size(); ++i) inst->limbs[i] = coeffs->items[i]->limbs[0];
                }
            }
        }
    }

    // parent now holds `child`: depth bookkeeping for the early refusal (exact check: check_shape)
    bool deepen(const Ref& parent, const Ref& child) {
        if (child && child->depth + 1 > parent->depth) parent->depth = child->depth + 1;
        if (parent->depth > 200) { why_ = "objects nested deeper than 200 levels"; return false; }
        return true;
    }
    bool deepen(const Ref& parent, const std::vector<Ref>& children) {
        for (const Ref& c : children) if (!deepen(parent, c)) return false;
        return true;
    }

    Ref run() {
        while (pos_ < n_) {
            // (a reference proof never has more than ~1100 objects on the stack -- one batch of 1000 list items plus a few levels.
            //  Nesting does NOT need stack depth: TUPLE1 pops one object and pushes one, so every container operation below also
            //  carries the depth of what it builds and refuses early; load() checks the finished graph exactly.)
            if (stack_.size() > 20000) { why_ = "unpickling stack deeper than 20000"; return Ref(); }
            const unsigned char opc = p_[pos_++];
            Ref a, b;
            std::vector<Ref> items;
            switch (opc) {
                case 0x80: if (!need(1)) return Ref(); if (p_[pos_++] > 5) { why_ = "pickle protocol"; return Ref(); } break;   // PROTO
                case 0x95: if (!need(8)) return Ref(); pos_ += 8; break;                                  // FRAME (the length is not needed)
                case 0x5d: stack_.push_back(mk(K_LIST)); break;                                           // EMPTY_LIST
                case 0x7d: stack_.push_back(mk(K_DICT)); break;                                           // EMPTY_DICT
                case 0x29: stack_.push_back(mk(K_TUPLE)); break;                                          // EMPTY_TUPLE
                case 0x94: if (stack_.empty()) { why_ = "MEMOIZE on an empty stack"; return Ref(); } memo_.push_back(stack_.back()); break;
                case 0x28: marks_.push_back(stack_.size()); break;                                        // MARK
                case 0x65:                                                                                // APPENDS
                    if (!pop_mark(items) || stack_.empty() || stack_.back()->kind != K_LIST) { why_ = "APPENDS"; return Ref(); }
                    stack_.back()->items.insert(stack_.back()->items.end(), items.begin(), items.end());
                    if (!deepen(stack_.back(), items)) return Ref();
                    break;
                case 0x61:                                                                                // APPEND
                    if (!pop(a) || stack_.empty() || stack_.back()->kind != K_LIST) { why_ = "APPEND"; return Ref(); }
                    stack_.back()->items.push_back(a);
                    if (!deepen(stack_.back(), a)) return Ref();
                    break;
                case 0x75:                                                                                // SETITEMS
                    if (!pop_mark(items) || (items.size() & 1) || stack_.empty() || stack_.back()->kind != K_DICT) { why_ = "SETITEMS"; return Ref(); }
                    stack_.back()->items.insert(stack_.back()->items.end(), items.begin(), items.end());
                    if (!deepen(stack_.back(), items)) return Ref();
                    break;
                case 0x73:                                                                                // SETITEM
                    if (!pop(b) || !pop(a) || stack_.empty() || stack_.back()->kind != K_DICT) { why_ = "SETITEM"; return Ref(); }
                    stack_.back()->items.push_back(a);
                    stack_.back()->items.push_back(b);
                    if (!deepen(stack_.back(), a) || !deepen(stack_.back(), b)) return Ref();
                    break;
                case 0x43: if (!blob(1, K_BYTES)) return Ref(); break;                                    // SHORT_BINBYTES
                case 0x42: if (!blob(4, K_BYTES)) return Ref(); break;                                    // BINBYTES
                case 0x8e: if (!blob(8, K_BYTES)) return Ref(); break;                                    // BINBYTES8
                case 0x8c: if (!blob(1, K_STR)) return Ref(); break;                                      // SHORT_BINUNICODE
                case 0x58: if (!blob(4, K_STR)) return Ref(); break;                                      // BINUNICODE
                case 0x4b: if (!need(1)) return Ref(); stack_.push_back(mk_int(le(1))); break;            // BININT1
                case 0x4d: if (!need(2)) return Ref(); stack_.push_back(mk_int(le(2))); break;            // BININT2
                case 0x4a: {                                                                              // BININT (signed)
                    if (!need(4)) return Ref();
                    const u64 v = le(4);
                    if (v >> 31) { why_ = "negative integer"; return Ref(); }
                    stack_.push_back(mk_int(v));
                    break;
                }
                case 0x8a: {                                                                              // LONG1
                    if (!need(1)) return Ref();
                    const size_t len = p_[pos_++];
                    if (!need(len)) return Ref();
                    if (len > 9 || (len == 9 && p_[pos_ + 8] != 0) || (len > 0 && len < 9 && (p_[pos_ + len - 1] & 0x80))) {
                        why_ = "integer outside [0, 2^64)";
                        return Ref();
                    }
                    stack_.push_back(mk_int(le(len > 8 ? 8 : len)));
                    if (len == 9) ++pos_;
                    break;
                }
                case 0x93:                                                                                // STACK_GLOBAL
                    if (!pop(b) || !pop(a) || a->kind != K_STR || b->kind != K_STR) { why_ = "STACK_GLOBAL"; return Ref(); }
                    stack_.push_back(mk_class(a, b));
                    break;
                case 0x81: {                                                                              // NEWOBJ: cls.__new__(cls, *args)
                    if (!pop(b) || !pop(a) || a->kind != K_CLASS || b->kind != K_TUPLE || !b->items.empty()) { why_ = "NEWOBJ"; return Ref(); }
                    Ref inst = mk(K_INSTANCE);
                    inst->cls = a;
                    instances_.push_back(inst);
                    stack_.push_back(inst);
                    break;
                }
                case 0x62:                                                                                // BUILD: instance.__dict__ = state
                    if (!pop(a) || a->kind != K_DICT || stack_.empty() || stack_.back()->kind != K_INSTANCE || stack_.back()->state) { why_ = "BUILD"; return Ref(); }
                    stack_.back()->state = a;
                    if (!deepen(stack_.back(), a)) return Ref();
                    tag(stack_.back());
                    break;
                case 0x85: case 0x86: case 0x87: {                                                        // TUPLE1 / 2 / 3
                    const size_t k = opc - 0x84;
                    if (stack_.size() < k || (!marks_.empty() && stack_.size() - k < marks_.back())) { why_ = "TUPLEn"; return Ref(); }
                    Ref t = mk(K_TUPLE);
                    t->items.assign(stack_.end() - k, stack_.end());
                    stack_.resize(stack_.size() - k);
                    if (!deepen(t, t->items)) return Ref();
                    stack_.push_back(t);
                    break;
                }
                case 0x74: {                                                                              // TUPLE
                    if (!pop_mark(items)) return Ref();
                    Ref t = mk(K_TUPLE);
                    t->items = items;
                    if (!deepen(t, t->items)) return Ref();
                    stack_.push_back(t);
                    break;
                }
                case 0x68: case 0x6a: {                                                                   // BINGET / LONG_BINGET
                    const size_t k = opc == 0x68 ? 1 : 4;
                    if (!need(k)) return Ref();
                    const u64 idx = le(k);
                    if (idx >= memo_.size()) { why_ = "memo index"; return Ref(); }
                    stack_.push_back(memo_[idx]);
                    break;
                }
                case 0x2e:                                                                                // STOP
                    if (stack_.size() != 1 || !marks_.empty()) { why_ = "STOP with a stack of the wrong depth"; return Ref(); }
                    // every instance must have been BUILT: the writer dereferences an instance's state.  (Until round 6 only the
                    // MEMOISED instances were checked here; a stream whose NEWOBJ is followed by neither MEMOIZE nor BUILD -- two byte
                    // mutations of a proof -- put an instance without state into the graph and the round-trip check of bfs_ps_loads
                    // crashed on it: found by tests/test_sanitized_parsers.py.)
                    for (const Ref& m : instances_)
                        if (!m->state) { why_ = "an instance without state"; return Ref(); }
                    return stack_.back();
                default: {
                    char buf[48];
                    snprintf(buf, sizeof buf, "opcode 0x%02x is not supported", opc);
                    why_ = buf;
                    return Ref();
                }
            }
        }
        why_ = "no STOP";
        return Ref();
    }
};

// ProofStream (ip.py:4-30): a list of objects, serialised as one pickle; handles are indices into an arena
struct Transcript {
    World world;
    std::vector<Ref> arena;    // every object ever created through the C ABI (handle = index + 1)
    std::vector<Ref> objects;  // the pushed objects, in order

    uint64_t add(const Ref& r) { arena.push_back(r); return arena.size(); }
    Ref get(uint64_t h) const { return (h >= 1 && h <= arena.size()) ? arena[h - 1] : Ref(); }

    // the whole stream is pickled incrementally (a proof asks for Fiat-Shamir randomness after almost every push, each time
    // over everything pushed so far); prefixes (the verifier's view) and streams of fewer than two objects take the plain route
    mutable Pickler stream{&world};
    mutable bool streaming = false;

    std::string serialize(size_t count) const {
        if (count >= objects.size() && objects.size() >= 2) {
            if (!streaming) { stream.stream_begin(); streaming = true; }
            for (size_t i = stream.stream_items(); i < objects.size(); ++i) stream.stream_item(objects[i]);
            return stream.stream_bytes();
        }
        Ref lst = mk(K_LIST);
        lst->items.assign(objects.begin(), objects.begin() + (count < objects.size() ? count : objects.size()));
        Pickler p(&world);
        return p.dumps(lst);
    }
    void fiat_shamir(size_t count, unsigned char* out, size_t num_bytes) const {
        if (!prefetched.empty()) {
            auto it = prefetched.find(count < objects.size() ? count : objects.size());
            if (it != prefetched.end() && it->second->num_bytes == num_bytes) {
                it->second->run();                                   // does the work here unless a helper has it (then waits for that helper)
                memcpy(out, it->second->out, num_bytes);
                return;
            }
        }
        std::string s = serialize(count);
        shake256(s.data(), s.size(), out, num_bytes);
    }

    // The verifier's side of Fiat-Shamir (ip.py:27-30): shake_256(pickle.dumps(objects[:read_index])) at ~20 read positions of a stream
    // that does not change any more.  The hashes are independent of each other (the pickle's frame header carries the prefix' length, so
    // no two share a sponge state) and of everything else the verifier does, so a stream read from bytes (bfs_ps_loads) can have them
    // computed ahead by the helper threads.  Jobs are OFFERS (helper_pool.hpp): whoever needs a result first computes it; a job owns
    // the objects of its prefix, so it outlives the stream if it has to.
    struct PrefixHash {
        std::shared_ptr<const std::vector<Ref>> all;                 // the stream's objects (one shared copy for every job of a stream) ...
        size_t count = 0;                                            // ... of which this job hashes the first `count`
        size_t num_bytes = 0;
        unsigned char out[64];
        std::atomic<int> state{0};                                   // 0 offered, 1 somebody is computing, 2 done
        void run() {
            int expected = 0;
            if (state.compare_exchange_strong(expected, 1)) {
                Pickler p(nullptr);                                  // (streams read from bytes hold no compact elements: no World needed)
                const std::string s = p.dumps_prefix(*all, count);
                shake256(s.data(), s.size(), out, num_bytes);
                state.store(2, std::memory_order_release);
                return;
            }
            while (state.load(std::memory_order_acquire) != 2) std::this_thread::yield();
        }
    };
    std::unordered_map<size_t, std::shared_ptr<PrefixHash>> prefetched;
    std::shared_ptr<const std::vector<Ref>> prefetch_objects;
    bool loaded_from_bytes = false;
    // hashes nobody asked for are withdrawn with the stream: a helper that gets to one later finds it taken and moves on (a verifier that
    // runs proof after proof would otherwise queue its next proof's hashes behind the last one's leftovers)
    ~Transcript() {
        for (auto& kv : prefetched) {
            int expected = 0;
            (void)kv.second->state.compare_exchange_strong(expected, 2);
        }
    }
    // offers the hashes over objects[:counts[i]] to the helper pool; returns how many were offered (0: no helpers, or not a loaded stream)
    size_t prefetch_fiat_shamir(const size_t* counts, size_t n, size_t num_bytes) {
        if (!loaded_from_bytes || num_bytes == 0 || num_bytes > 64) return 0;
        HelperPool* pool = HelperPool::get();
        if (pool == nullptr) return 0;
        std::vector<std::function<void()>> work;
        if (!prefetch_objects) prefetch_objects = std::make_shared<const std::vector<Ref>>(objects);      // (a loaded stream does not change)
        for (size_t i = 0; i < n; ++i) {
            const size_t k = counts[i] < objects.size() ? counts[i] : objects.size();
            if (prefetched.count(k)) continue;
            std::shared_ptr<PrefixHash> job = std::make_shared<PrefixHash>();
            job->all = prefetch_objects;
            job->count = k;
            job->num_bytes = num_bytes;
            prefetched[k] = job;
            work.push_back([job] { int s = job->state.load(); if (s == 0) job->run(); });
        }
        const size_t offered = work.size();
        if (offered) pool->submit(std::move(work));
        return offered;
    }

    // Fiat-Shamir over a stream whose LAST object is a 64-byte digest that is still being computed (a Merkle root on its way from the
    // GPU).  Every challenge hashes the whole transcript (tens of KB, ip.py:21-25), but the length of the final pickle -- hence its
    // frame header -- and everything before the digest's payload are known beforehand: speculate() pushes a placeholder, pickles,
    // and absorbs all SHAKE blocks in front of the payload while the kernel runs; resolve() fills the digest in and finishes.
    struct Speculation {
        std::string bytes;
        size_t payload = 0, absorbed = 0;
        uint64_t sponge[25];
        Ref node;
        bool active = false;
    };
    bool speculate(Speculation& sp) {
        unsigned char sentinel[64];
        for (int i = 0; i < 64; ++i) sentinel[i] = (unsigned char)(0xC3 ^ (i * 37));
        sp.node = mk_bytes(sentinel, 64);
        objects.push_back(sp.node);
        sp.active = false;
        if (objects.size() < 2) return false;                       // plain route in serialize(): no open form to patch
        sp.bytes = serialize(objects.size());
        // the payload sits in the last ~80 bytes: SHORT_BINBYTES 64 <payload> MEMOIZE [APPENDS] STOP
        const size_t from = sp.bytes.size() > 96 ? sp.bytes.size() - 96 : 0;
        const size_t at = sp.bytes.find(std::string((const char*)sentinel, 64), from);
        if (at == std::string::npos) return false;
        sp.payload = at;
        sp.absorbed = shake256_absorb_blocks(sp.bytes.data(), at, sp.sponge);
        sp.active = true;
        return true;
    }
    // The same idea for a RUN of digests (the commit phase of FRI pushes one root per round and asks for a challenge after each,
    // fri.py:108-120): the pickles of "stream + k digests" for k = 1..count are all made up front with placeholder payloads, and
    // helper threads absorb each one's blocks in front of the first placeholder (helper_pool.hpp) -- tens of KB per round that
    // would otherwise be hashed on the proving thread between two kernel launches.  next() then pushes the real digest, fills the
    // k payloads in and finishes the sponge over the last few hundred bytes.
    struct Lookahead {
        // A job is an offer to the helper threads, not a dependency on them: whoever gets to it first -- a helper, or the proving
        // thread when it needs the result (or is tearing the look-ahead down) -- claims it with a compare-and-swap and does the
        // work; the only wait left is for a helper that is already running the job.  The lambdas in the pool's queue share
        // ownership of the job, so one that is dequeued late (after the proof) finds a claimed job in live memory and returns.
        struct Job {
            std::string tail;                        // the pickle of "stream + k digests" from offset `from` on
            unsigned char watched[9];                // ... and its nine bytes at `watch` (the frame header in front depends on k)
            const std::string* base = nullptr;
            size_t from = 0, watch = 0;
            uint64_t sponge[25];
            enum { PENDING = 0, RUNNING = 1, DONE = 2 };
            std::atomic<int> state{PENDING};
            bool by_helper = false;                  // the lambda in the pool's queue has been consumed (written before DONE is stored)
            bool claim() {
                int expected = PENDING;
                return state.compare_exchange_strong(expected, RUNNING, std::memory_order_acq_rel);
            }
            void absorb() {
                shake256_absorb_blocks_patched(base->data(), from, sponge, watch, watched, 9);
            }
            // the result is needed now: do it here unless a helper already has it in hand
            void finish_here() {
                if (claim()) {
                    absorb();
                    state.store(DONE, std::memory_order_release);
                    return;
                }
                for (unsigned spins = 0; state.load(std::memory_order_acquire) != DONE; ++spins)
                    if (spins > 4096) std::this_thread::yield();          // (a RUNNING helper that lost its core to somebody else)
            }
            // nobody needs the result any more: make sure no helper touches `base` from here on
            void retire() {
                if (claim()) { state.store(DONE, std::memory_order_release); return; }
                while (state.load(std::memory_order_acquire) != DONE) std::this_thread::yield();
            }
        };
        std::string base;                            // the whole pickle for k = 1: every job's prefix, up to the nine watched bytes
        std::vector<std::shared_ptr<Job>> jobs;      // jobs[k - 1]: the stream followed by k digests
        std::vector<size_t> holes;                   // offset of the k-th digest's payload, the same in every pickle that holds it
        std::vector<std::string> digests;
        size_t from = 0, base_objects = 0;
        bool active = false;
        void wait_all() {
            for (auto& j : jobs) j->retire();
        }
        // finished jobs (and the base) are kept per thread for the next proof: their strings keep their capacity.  Only jobs whose
        // queued lambda has run are recycled -- one that was finished here may still be referenced by a lambda waiting in the pool.
        struct Spare { std::vector<std::shared_ptr<Job>> jobs; std::string base; };
        static Spare& spare() { static thread_local Spare s; return s; }
        static std::shared_ptr<Job> take() {
            auto& s = spare().jobs;
            if (s.empty()) return std::make_shared<Job>();
            std::shared_ptr<Job> j = std::move(s.back());
            s.pop_back();
            j->by_helper = false;
            j->state.store(Job::PENDING, std::memory_order_relaxed);
            return j;
        }
        Lookahead() { base.swap(spare().base); }
        ~Lookahead() {                               // the helpers read `base` and write into the jobs
            wait_all();
            for (auto& j : jobs)
                if (j->by_helper && spare().jobs.size() < 64) spare().jobs.push_back(std::move(j));
            base.swap(spare().base);
        }
    };
    static void lookahead_sentinel(size_t k, unsigned char out[64]) {
        for (int i = 0; i < 64; ++i) out[i] = (unsigned char)(0xC3 ^ (i * 37) ^ (k * 101));
    }
    // false: nothing started (short stream, no helpers, or the pickles did not come out the expected shape) -- use speculate()
    bool lookahead_begin(Lookahead& la, size_t count, size_t min_bytes = 4096) {
        la.active = false;
        if (count == 0 || objects.size() < 2) return false;
        HelperPool* pool = HelperPool::get();
        if (pool == nullptr) return false;
        if (!streaming) { stream.stream_begin(); streaming = true; }
        for (size_t i = stream.stream_items(); i < objects.size(); ++i) stream.stream_item(objects[i]);   // bring the open form up to date
        if (stream.stream_size() < min_bytes) return false;
        const Pickler::StreamMark mark = stream.stream_mark();
        const size_t watch = stream.stream_open_frame();
        std::vector<Ref> tentative;
        bool ok = true;
        for (size_t k = 1; k <= count && ok; ++k) {
            unsigned char sentinel[64];
            lookahead_sentinel(k, sentinel);
            tentative.push_back(mk_bytes(sentinel, 64));
            stream.stream_item(tentative.back());
            std::shared_ptr<Lookahead::Job> job = Lookahead::take();
            if (k == 1) {                            // the one whole copy; everything in front of the first payload's block is shared
                stream.stream_bytes_into(la.base);
                const size_t near = la.base.size() > 96 ? la.base.size() - 96 : 0;
                const size_t at = la.base.find(std::string((const char*)sentinel, 64), near);
                if (at == std::string::npos) { ok = false; break; }
                la.holes.push_back(at);
                la.from = at - at % 136;
            }
            stream.stream_tail_into(job->tail, la.from, watch, job->watched);
            if (k > 1) {
                const size_t near = job->tail.size() > 96 ? job->tail.size() - 96 : 0;
                const size_t at = job->tail.find(std::string((const char*)sentinel, 64), near);
                if (at == std::string::npos) { ok = false; break; }
                la.holes.push_back(la.from + at);
            }
            for (size_t j = 1; j <= k && ok; ++j) {                           // the payloads sit where they sat
                lookahead_sentinel(j, sentinel);
                const size_t at = la.holes[j - 1] - la.from;
                if (at + 64 > job->tail.size() || memcmp(&job->tail[at], sentinel, 64) != 0) ok = false;
            }
            job->base = &la.base;
            job->from = la.from;
            job->watch = watch;
            la.jobs.push_back(std::move(job));
        }
        stream.stream_rollback(mark, tentative);
        if (!ok) { la.jobs.clear(); la.holes.clear(); return false; }
        std::vector<std::function<void()>> work;
        for (auto& j : la.jobs) {
            std::shared_ptr<Lookahead::Job> job = j;
            work.push_back([job] {
                if (!job->claim()) return;           // the proving thread got there first (or the look-ahead is gone)
                job->absorb();
                job->by_helper = true;
                job->state.store(Lookahead::Job::DONE, std::memory_order_release);
            });
        }
        pool->submit(std::move(work));
        la.base_objects = objects.size();
        la.active = true;
        return true;
    }
    // push the next digest; out != nullptr: the Fiat-Shamir bytes over the stream including it
    void lookahead_next(Lookahead& la, const unsigned char digest[64], unsigned char* out, size_t num_bytes) {
        objects.push_back(mk_bytes(digest, 64));
        const size_t k = la.digests.size() + 1;
        la.digests.emplace_back((const char*)digest, 64);
        if (out == nullptr) return;
        if (!la.active || k > la.jobs.size() || objects.size() != la.base_objects + k) { fiat_shamir(objects.size(), out, num_bytes); return; }
        Lookahead::Job& job = *la.jobs[k - 1];
        for (size_t j = 0; j < k; ++j) memcpy(&job.tail[la.holes[j] - la.from], la.digests[j].data(), 64);
        job.finish_here();
        shake256(job.tail.data(), job.tail.size(), out, num_bytes, job.sponge, 0);
    }

    void resolve(Speculation& sp, const unsigned char digest[64], unsigned char* out, size_t num_bytes) {
        sp.node->set_bytes(digest, 64);
        if (!sp.active) { fiat_shamir(objects.size(), out, num_bytes); return; }
        memcpy(&sp.bytes[sp.payload], digest, 64);
        stream.stream_patch(sp.payload, digest, 64);
        shake256(sp.bytes.data(), sp.bytes.size(), out, num_bytes, sp.sponge, sp.absorbed);
        sp.active = false;
    }
};

// BaseField.sample (algebra.py:138-142): big-endian bytes -> integer mod p
inline u64 sample_base(const unsigned char* b, size_t len) {
    // eight bytes at a time: acc <- acc * 2^64 + word, 2^64 = 2^32 - 1 (mod p)   (was one 128-bit division per byte)
    u64 acc = 0;
    size_t i = 0;
    for (; i < len % 8; ++i) acc = (acc << 8) | b[i];
    for (; i < len; i += 8) {
        u64 w = 0;
        for (int k = 0; k < 8; ++k) w = (w << 8) | b[i + k];
        acc = gl_add(gl_mul(acc % GL_P, 0xFFFFFFFFULL), w % GL_P);
    }
    return acc % GL_P;
}
// ExtensionField.sample (extension_field.py:100-111): three chunks of len//3 bytes
inline Xfe sample_xfe(const unsigned char* b, size_t len) {
    size_t c = len / 3;
    return Xfe{{sample_base(b, c), sample_base(b + c, c), sample_base(b + 2 * c, c)}};
}

}  // namespace rp
}  // namespace bfs
