// vm.cpp -- the Brainfuck virtual machine and its execution trace, host code:
//   VirtualMachine.simulate      /root/reference/code/vm.py:172-306   processor / instruction / input / output matrices
//   MemoryTable.derive_matrix    memory_table.py:20-38                memory matrix (sorted by address, dummy rows for clock jumps)
// The reference builds ~10^6 element objects for a 37 000-cycle program (seconds of interpreter time before the prover starts);
// here the machine runs natively and the matrices come back as integer arrays.  The Python mirror wraps them in lazily
// materialised matrices (vm.py: LazyTraceMatrix), so the call surface stays "list of rows of elements".
//
// Object identity: in the reference a memory cell holds an element OBJECT and the memory-value register is whatever object sits
// in the current cell (vm.py:266-292); the input / output matrices hold those same objects.  Identity reaches the proof through
// the first term of the running evaluations (processor_table.py:390-404) because pickle memoises by identity.  The trace therefore
// carries, next to every memory value, the id of the object that held it: 0 = the register's initial zero (vm.py:188), 1 = the
// shared zero that untouched cells read as, k > 1 = an object created by `+`, `-` or `,`.
#include <algorithm>
#include <unordered_map>
#include <vector>

#include "../../include/bfstark.h"
#include "gl.hpp"
#include "runtime.hpp"

namespace bfs {

struct VmTrace {
    std::vector<u64> processor;        // rows x 7: clk ip ci ni mp mv mvi
    std::vector<u64> processor_ids;    // rows: object id of the mv entry
    std::vector<u64> instruction;      // rows x 3: ip ci ni, sorted by ip (stable)
    std::vector<u64> memory;           // rows x 4: clk mp mv dummy
    std::vector<u64> input, input_ids, output, output_ids;
    u64 objects = 1;                   // ids handed out so far (0 and 1 are the two zeros)
};

}  // namespace bfs

using namespace bfs;

// rows 0..m-1 in stable order of key(row): a counting sort when the keys are small (instruction addresses, memory pointers of
// ordinary programs), std::stable_sort otherwise (a pointer that wrapped below zero is p - 1)
template <class Key>
static std::vector<u32> stable_order(size_t m, u64 max_key, Key key) {
    std::vector<u32> order(m);
    if (max_key < (1u << 22) && max_key < 8 * m + 1024) {
        std::vector<u32> start(max_key + 2, 0);
        for (size_t k = 0; k < m; ++k) ++start[key(k) + 1];
        for (size_t v = 0; v <= max_key; ++v) start[v + 1] += start[v];
        for (size_t k = 0; k < m; ++k) order[start[key(k)]++] = (u32)k;
    } else {
        for (size_t k = 0; k < m; ++k) order[k] = (u32)k;
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return key(a) < key(b); });
    }
    return order;
}

extern "C" {

int bfs_vm_trace_new(const uint64_t* program, size_t n, const uint32_t* input, size_t n_input, uint64_t max_cycles, void** trace) {
    if (n == 0) { set_error("bfs_vm_trace_new: empty program"); return BFS_ERR_BAD_ARG; }
    // the trace grows by ~130 bytes per cycle and `-[-]` counts down from p - 1: never run unbounded by default
    if (max_cycles == 0) max_cycles = BFS_VM_DEFAULT_MAX_CYCLES;
    struct Cell { u64 value, id; };
    // cells 0 .. 2^16 - 1 in a flat array (a never-written cell reads as value 0 / the shared zero object, id 1), the rest -- a pointer
    // that ran far to the right, or wrapped below zero -- in a hash map
    constexpr u64 FLAT = 1u << 16;
    std::vector<Cell> flat(FLAT, Cell{0, 1});
    std::unordered_map<u64, Cell> far;
    VmTrace* t = new VmTrace();
    u64 clk = 0, ip = 0, mp = 0;
    u64 ci = program[0] % GL_P, ni = n > 1 ? program[1] % GL_P : 0;
    Cell mv{0, 0};
    size_t input_counter = 0;
    std::vector<u64>& in = t->instruction;
    t->processor.reserve(7 << 16); t->processor_ids.reserve(1 << 16); in.reserve(3 * (n + (1 << 16)));
    for (size_t i = 0; i + 1 < n; ++i) { in.push_back(i); in.push_back(program[i] % GL_P); in.push_back(program[i + 1] % GL_P); }
    in.push_back(n - 1); in.push_back(program[n - 1] % GL_P); in.push_back(0);
    auto row = [&]() {
        const u64 r[7] = {clk, ip, ci, ni, mp, mv.value, 0};
        t->processor.insert(t->processor.end(), r, r + 7);
        t->processor_ids.push_back(mv.id);
        in.push_back(ip); in.push_back(ci); in.push_back(ni);
    };
    auto cell = [&](u64 address) -> Cell {
        if (address < FLAT) return flat[address];
        auto it = far.find(address);
        return it == far.end() ? Cell{0, 1} : it->second;
    };
    auto store = [&](u64 address, Cell c) {
        if (address < FLAT) flat[address] = c;
        else far[address] = c;
    };
    while (ip < n) {
        if (clk >= max_cycles) { delete t; set_error("program runs for more than %llu cycles", (unsigned long long)max_cycles); return BFS_ERR_BAD_ARG; }
        row();
        switch (ci) {
            case '[': ip = mv.value == 0 ? program[ip + 1] % GL_P : ip + 2; break;
            case ']': ip = mv.value != 0 ? program[ip + 1] % GL_P : ip + 2; break;
            case '<': ip += 1; mp = gl_sub(mp, 1); break;
            case '>': ip += 1; mp = gl_add(mp, 1); break;
            case '+': ip += 1; store(mp, Cell{gl_add(cell(mp).value, 1), ++t->objects}); break;
            case '-': ip += 1; store(mp, Cell{gl_sub(cell(mp).value, 1), ++t->objects}); break;
            case '.': {
                ip += 1;
                const Cell c = cell(mp);
                t->output.push_back(c.value); t->output_ids.push_back(c.id);
                break;
            }
            case ',': {
                ip += 1;
                if (input_counter >= n_input) { delete t; set_error("program reads more input symbols than were supplied"); return BFS_ERR_BAD_ARG; }
                const Cell c{(u64)input[input_counter++] % GL_P, ++t->objects};
                store(mp, c);
                t->input.push_back(c.value); t->input_ids.push_back(c.id);
                break;
            }
            default:
                delete t;
                set_error("unrecognized instruction at %llu: %llu", (unsigned long long)ip, (unsigned long long)ci);
                return BFS_ERR_BAD_ARG;
        }
        clk += 1;
        ci = ip < n ? program[ip] % GL_P : 0;
        ni = ip + 1 < n ? program[ip + 1] % GL_P : 0;
        mv = cell(mp);
    }
    row();
    // memory-value inverses, all at once (Montgomery's trick; zero stays zero)
    const size_t rows = t->processor_ids.size();
    {
        std::vector<u64> prefix(rows);
        u64 running = 1;
        for (size_t r = 0; r < rows; ++r) {
            prefix[r] = running;
            const u64 v = t->processor[7 * r + 5];
            if (v) running = gl_mul(running, v);
        }
        u64 inv = gl_inv(running);
        for (size_t r = rows; r-- > 0;) {
            const u64 v = t->processor[7 * r + 5];
            if (v) { t->processor[7 * r + 6] = gl_mul(inv, prefix[r]); inv = gl_mul(inv, v); }
        }
    }
    // instruction matrix: stable sort by address (vm.py:302)
    {
        const size_t m = in.size() / 3;
        u64 max_key = 0;
        for (size_t k = 0; k < m; ++k) max_key = in[3 * k] > max_key ? in[3 * k] : max_key;
        const std::vector<u32> order = stable_order(m, max_key, [&](size_t k) { return in[3 * k]; });
        std::vector<u64> sorted(in.size());
        for (size_t k = 0; k < m; ++k) for (int j = 0; j < 3; ++j) sorted[3 * k + j] = in[3 * (size_t)order[k] + j];
        in.swap(sorted);
    }
    // memory matrix (memory_table.py:20-38): non-padding rows sorted by address (stable), dummy rows where the clock jumps
    {
        std::vector<u32> live;
        u64 max_key = 0;
        for (size_t r = 0; r < rows; ++r)
            if (t->processor[7 * r + 2] != 0) {
                live.push_back((u32)r);
                const u64 a = t->processor[7 * r + 4];
                max_key = a > max_key ? a : max_key;
            }
        const std::vector<u32> pos = stable_order(live.size(), max_key, [&](size_t k) { return t->processor[7 * (size_t)live[k] + 4]; });
        std::vector<u32> order(live.size());
        for (size_t k = 0; k < live.size(); ++k) order[k] = live[pos[k]];
        // size first (a dummy row for every clock value skipped between two consecutive visits of an address), then one pass of plain
        // stores: 111 546 rows for 37 254 cycles, and per-row vector inserts were the largest item of the whole call
        size_t total = order.size();
        for (size_t k = 0; k + 1 < order.size(); ++k) {
            const u64* p = &t->processor[7 * (size_t)order[k]];
            const u64* q = &t->processor[7 * (size_t)order[k + 1]];
            if (q[4] == p[4]) total += (size_t)(gl_sub(q[0], p[0]) - 1);      // clocks grow along the visits of one address
        }
        std::vector<u64>& mm = t->memory;
        mm.resize(4 * total);
        u64* w = mm.data();
        for (size_t k = 0; k < order.size(); ++k) {
            const u64* p = &t->processor[7 * (size_t)order[k]];
            w[0] = p[0]; w[1] = p[4]; w[2] = p[5]; w[3] = 0;
            w += 4;
            if (k + 1 < order.size()) {
                const u64* q = &t->processor[7 * (size_t)order[k + 1]];
                if (q[4] == p[4]) {
                    u64 c = p[0];
                    while (gl_add(c, 1) != q[0]) {
                        c = gl_add(c, 1);
                        w[0] = c; w[1] = p[4]; w[2] = p[5]; w[3] = 1;
                        w += 4;
                    }
                }
            }
        }
    }
    *trace = t;
    return BFS_OK;
}

void bfs_vm_trace_free(void* trace) { delete (VmTrace*)trace; }

// which: 0 processor (rows x 7), 1 memory (rows x 4), 2 instruction (rows x 3), 3 input, 4 output,
//        5 object ids of the processor's memory-value column, 6 / 7 object ids of the input / output symbols
static const std::vector<u64>* vm_part(const VmTrace* t, int which) {
    switch (which) {
        case 0: return &t->processor;
        case 1: return &t->memory;
        case 2: return &t->instruction;
        case 3: return &t->input;
        case 4: return &t->output;
        case 5: return &t->processor_ids;
        case 6: return &t->input_ids;
        case 7: return &t->output_ids;
    }
    return nullptr;
}

int bfs_vm_trace_size(void* trace, int which, size_t* words) {
    const std::vector<u64>* v = vm_part((const VmTrace*)trace, which);
    if (!v) { set_error("bfs_vm_trace_size: part %d", which); return BFS_ERR_BAD_ARG; }
    *words = v->size();
    return BFS_OK;
}

int bfs_vm_trace_copy(void* trace, int which, uint64_t* out) {
    const std::vector<u64>* v = vm_part((const VmTrace*)trace, which);
    if (!v) { set_error("bfs_vm_trace_copy: part %d", which); return BFS_ERR_BAD_ARG; }
    std::copy(v->begin(), v->end(), out);
    return BFS_OK;
}

}  // extern "C"
