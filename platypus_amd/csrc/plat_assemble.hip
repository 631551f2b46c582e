// plat_assemble.hip -- coloured de-Bruijn local assembler on the device.
//
// Reference: src/cython/assembler.pyx (assembleReadsAndDetectVariants :1429-1476, graph construction :668-827,
// :1295-1387, bubble walk :1027-1177, variant extraction :1196-1291).  The reference builds the graph
// SEQUENTIALLY and its result depends on insertion order (node order = first insertion, edge slots = first
// insertion, at most 4 out-edges, LIFO path stack with ">20" aborts).  The device build is parallel over k-mer
// occurrences and reconstructs exactly that order from TICKETS: every AddEdge event e has the ticket the
// sequential loader would give it (reference edges first, then reads in buffer order), and
//   * a node's identity is its k bytes; its weight is the sum over all touches, its colours the OR;
//   * a node's position is that of its first touch when that touch comes from the reference (the reference is
//     loaded first, assembler.pyx:1449), else -1;
//   * a node keeps the (up to) four successors with the smallest first-occurrence tickets, in ticket order
//     (assembler.pyx:813-824: a fifth distinct successor is silently dropped);
//   * bubble starts are visited in allNodes order, which for REF_AND_READ nodes is increasing position.
// One workgroup assembles one region; everything an AddEdge event touches lives in the workgroup's LDS (the "LDS path"):
//   * the k-mer hash table (ASM_LDS_SLOTS slots: byte offsets of representative occurrences while the k-mers are inserted, then
//     node id << 18 | offset), one word per node for its first touch and one for its colours + the summed weight of its
//     first-claimed successor slot (ASM_LDS_NODES nodes), and the region's reference bytes (ASM_REF_CACHE);
//   * phase A inserts the reference's k-mers first, so a k-mer the reference holds is represented by a reference occurrence
//     and a read's k-mer is compared with its representative in LDS; the reads' bases and qualities are loaded as 256-byte
//     windows, one aligned dword per lane, and every lane gathers its edge's k+1 bytes from the other lanes (per-lane
//     unaligned 64-bit loads were bound by the texture addresser); every edge leaves one word (table slot, weight, appended
//     base) so that phase C neither hashes nor compares: it is a loop over tickets;
//   * node ids follow the reference (bitmap of representative positions + popcount ranks): reads are sorted by position,
//     so the few global words an event still updates (successor slots other than a node's first-claimed one) are neighbours
//     in memory and stay in the L2;
//   * which node a successor slot leads to is looked up once per slot (phase D), and first tickets -- which only order the
//     successors of a node that has several -- are collected for those nodes alone in a second pass over the event words.
// Measured on BASELINE config 3 (2000 tiles, PLAT_ASM_TIMING=1 prints the split): 84 k regions/s with the table in LDS but
// three global atomics and two hash look-ups per event; 123 k with reference-first insertion + LDS reference + one look-up per
// event; 141 k with the window loads; 174 k with one global atomic per event; 183 k with ordered ids; 214 k with the
// first-claimed slot's weight in LDS; 246 k with the slot words kept clean between regions.
// Round 4 (the fused pass, see "LDS path, fused" below): k-mers and events of the reads in one pass 278 k; the edges that pass the
// quality rule compacted before the probes 322 k; phases D-G on LDS (edge word per node in the table's space, walk stacks, path
// elements, one thread per finished path) and the claimed slots' end nodes in a dense array 336 k (768 threads: at 1024 the kernel
// spilled 139-175 registers to slots that live in HBM); phase boundaries without the L1 invalidate where only plainly stored words
// are read back (asm_sync_wg) 404 k; per-thread values re-derived per phase + no machine LICM (32 spilled registers) and 1024
// threads again 449-458 k; colour bits by unconditional LDS atomics, the walk's path array, used slots counted in phase D, one L1 invalidate per region: 545 k, 1.05 GB of
// HBM traffic per 2000 tiles (round 3: 252 k, 4.76 GB).  A region with more distinct k-mers than ASM_LDS_LIMIT (deep or very divergent data),
// k > 15, or a reference / read blob beyond 2^18 bytes is done with table, node words and successor lists in the workgroup's
// slice of a global scratch buffer (the "global path", the round-1 code).
#include "plat_internal.hpp"
#include <type_traits>

namespace plat {

constexpr int ASM_MAX_SUCC = 8;        // distinct successor bytes tracked per node before the "first four" cut
constexpr int ASM_MAX_TASKS = 512;     // bubble-start (node, edge) pairs per region
constexpr int ASM_POOL = 1 << 20;      // path elements per region, shared by its tasks (bump-allocated)
constexpr int ASM_MAX_FIN = 21;        // finished paths per task before the reference aborts (assembler.pyx:1052)
constexpr int ASM_THREADS = 1024;      // threads per workgroup (one workgroup per CU: the graph takes the LDS) = 128 registers per lane: see the Makefile's note on spills
constexpr int ASM_LDS_SLOTS = 16384;   // k-mer table in LDS: 64 KB
constexpr int ASM_LDS_NODES = 11264;   // per-node first-touch codes and (weight | colour << 30) words in LDS: 2 x 44 KB
constexpr int ASM_LDS_LIMIT = ASM_LDS_NODES - ASM_THREADS;   // distinct k-mers the LDS path takes (threads in flight may overshoot by one each)
constexpr int ASM_REF_CACHE = 7552;    // bytes of the region's reference kept in LDS (k-mers of reads are compared with their representative, which is a
                                       // reference k-mer whenever the reference holds one: the reference's k-mers are inserted first)
constexpr int ASM_LDS_BYTES = (ASM_LDS_SLOTS + 2 * ASM_LDS_NODES) * 4 + ASM_REF_CACHE;
constexpr int ASM_PATH = 64;           // nodes of the walk's current path kept as an array (16 bits each) for the cycle check; deeper paths walk the parent links
constexpr int ASM_STK = 28;            // pending path elements per bubble walk (the reference aborts a walk with more than 20, assembler.pyx:1052-1057)
constexpr int ASM_OFF_BITS = 18;       // LDS path: a table slot packs (node id << 18 | byte offset of the representative)
static_assert(ASM_LDS_NODES < ASM_LDS_SLOTS * 3 / 4, "the table must stay sparse");

struct AsmParams {
    int kmer, min_qual, min_weight, no_cycles, max_vars, blob_per_region;
    long long scratch_per_block;       // bytes
    int cap;                           // hash slots per region (power of two)
    int max_pos;                       // max k-mer occurrences per region (dense node capacity)
    int timing;                        // PLAT_ASM_TIMING: phase timers on
    int debug;                         // PLAT_ASM_DEBUG: 1 = no events in the fused reads pass, 2 = no table probes either (measurement only, results are garbage);
                                       // 4 = walk stacks in the slice and path elements in the global arena, 8 = variants extracted by one thread (tests: same results)
    int fused;                         // LDS path: k-mers and AddEdge events of the reads in ONE pass (round 4; PLAT_ASM_FUSED=0: rounds 2-3's passes)
};

struct AsmNodeE { int end[4]; int w[4]; int n; };   // finalised out-edges

// scratch layout per workgroup (all int32 unless noted)
struct AsmScratch {
    int* key;            // [cap]      byte offset of a representative occurrence, -1 empty
    int* slot_id;        // [cap]      dense node id
    unsigned* first;     // [max_pos]  min touch code (2*ticket + isEnd)
    int* weight;         // [max_pos]
    int* colour;         // [max_pos]
    int* rep;            // [max_pos]  representative byte offset
    unsigned char* succ_c;   // [max_pos][8]
    unsigned* succ_t;    // [max_pos][8]
    int* succ_w;         // [max_pos][8]
    int* succ_n;         // [max_pos][8]
    unsigned long long* succ_cw;   // [max_pos][8]  LDS path: events << 32 | weight of the successor slot
    AsmNodeE* edges;     // [max_pos]
    char* dfs;           // [max_pos]
    int* ref_node;       // [refLen]
    int* read_base;      // [nReads+1] ticket base per read
    int* task_node;      // [MAX_TASKS]
    int* task_edge;      // [MAX_TASKS]
    int* task_nfin;      // [MAX_TASKS]  (-1 aborted)
    int* task_fin;       // [MAX_TASKS][MAX_FIN] arena index of the last element
    int* arena;          // [POOL][3] node, parent, depth (parent = pool index)
    int* var_task_off;   // [MAX_TASKS+1]
    int* stack;          // [max_pos*2] iterative DFS stack for the cycle check
};

__host__ __device__ inline size_t asm_align(size_t x) { return (x + 15) & ~(size_t)15; }

// ints of the slice's `stack`: the DFS stack of the cycle check (2 per node), later the walks' stacks and the list of finished paths
__host__ __device__ inline size_t asm_stack_ints(int max_pos) {
    const size_t a = (size_t)max_pos * 2, b = (size_t)ASM_MAX_TASKS * (2 * ASM_MAX_FIN + ASM_STK) + 64;
    return a > b ? a : b;
}
__host__ __device__ inline size_t asm_scratch_bytes(int cap, int max_pos, int max_ref, int max_reads) {
    size_t b = 0;
    b += asm_align((size_t)cap * 4) * 2;
    b += asm_align((size_t)max_pos * 4) * 4;
    b += asm_align((size_t)max_pos * ASM_MAX_SUCC);
    b += asm_align((size_t)max_pos * ASM_MAX_SUCC * 4) * 3;
    b += asm_align((size_t)max_pos * ASM_MAX_SUCC * 8);
    b += asm_align((size_t)max_pos * sizeof(AsmNodeE));
    b += asm_align((size_t)max_pos);
    b += asm_align((size_t)(max_ref + 1) * 4);
    b += asm_align((size_t)(max_reads + 1) * 4);
    b += asm_align((size_t)ASM_MAX_TASKS * 4) * 3;
    b += asm_align((size_t)ASM_MAX_TASKS * ASM_MAX_FIN * 4);
    b += asm_align((size_t)ASM_POOL * 3 * 4);
    b += asm_align((size_t)(ASM_MAX_TASKS + 1) * 4);
    b += asm_align(asm_stack_ints(max_pos) * 4);
    return b;
}

__device__ inline AsmScratch asm_carve(char* p, int cap, int max_pos, int max_ref, int max_reads) {
    AsmScratch s;
    auto take = [&](size_t bytes) { char* r = p; p += asm_align(bytes); return r; };
    s.key = (int*)take((size_t)cap * 4);
    s.slot_id = (int*)take((size_t)cap * 4);
    s.first = (unsigned*)take((size_t)max_pos * 4);
    s.weight = (int*)take((size_t)max_pos * 4);
    s.colour = (int*)take((size_t)max_pos * 4);
    s.rep = (int*)take((size_t)max_pos * 4);
    s.succ_c = (unsigned char*)take((size_t)max_pos * ASM_MAX_SUCC);
    s.succ_t = (unsigned*)take((size_t)max_pos * ASM_MAX_SUCC * 4);
    s.succ_w = (int*)take((size_t)max_pos * ASM_MAX_SUCC * 4);
    s.succ_n = (int*)take((size_t)max_pos * ASM_MAX_SUCC * 4);
    s.succ_cw = (unsigned long long*)take((size_t)max_pos * ASM_MAX_SUCC * 8);
    s.edges = (AsmNodeE*)take((size_t)max_pos * sizeof(AsmNodeE));
    s.dfs = (char*)take((size_t)max_pos);
    s.ref_node = (int*)take((size_t)(max_ref + 1) * 4);
    s.read_base = (int*)take((size_t)(max_reads + 1) * 4);
    s.task_node = (int*)take((size_t)ASM_MAX_TASKS * 4);
    s.task_edge = (int*)take((size_t)ASM_MAX_TASKS * 4);
    s.task_nfin = (int*)take((size_t)ASM_MAX_TASKS * 4);
    s.task_fin = (int*)take((size_t)ASM_MAX_TASKS * ASM_MAX_FIN * 4);
    s.arena = (int*)take((size_t)ASM_POOL * 3 * 4);
    s.var_task_off = (int*)take((size_t)(ASM_MAX_TASKS + 1) * 4);
    s.stack = (int*)take(asm_stack_ints(max_pos) * 4);
    return s;
}

// Sequence bytes are read eight at a time (unaligned 64-bit loads; every blob is followed by PLAT_BLOB_PAD readable bytes):
// measured, the kernel was bound by the rate of byte loads once its atomics had moved to the LDS.
typedef unsigned long long asm_u64 __attribute__((aligned(1)));
__device__ __forceinline__ unsigned long long asm_ld8(const uint8_t* p) { return *(const asm_u64*)p; }
__device__ __forceinline__ unsigned long long asm_tailmask(int nbytes) { return nbytes >= 8 ? ~0ull : ((1ull << (8 * nbytes)) - 1ull); }
__device__ __forceinline__ unsigned asm_hash(const uint8_t* p, int k) {
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < k; i += 8) {
        const unsigned long long w = asm_ld8(p + i) & asm_tailmask(k - i);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 29;
    }
    return (unsigned)(h ^ (h >> 32));
}
__device__ __forceinline__ bool asm_eq(const uint8_t* a, const uint8_t* b, int k) {
    for (int i = 0; i < k; i += 8)
        if ((asm_ld8(a + i) ^ asm_ld8(b + i)) & asm_tailmask(k - i)) return false;
    return true;
}
// byte pointer of an occurrence: offsets < 0x40000000 index the reference, others the read blob
__device__ __forceinline__ const uint8_t* asm_ptr(const uint8_t* ref, const uint8_t* rseq, int off) {
    return off >= 0x40000000 ? rseq + (off - 0x40000000) : ref + off;
}

// find (or, if insert, create) the slot of the k-mer starting at byte offset `off`
__device__ inline int asm_slot(AsmScratch& S, const uint8_t* ref, const uint8_t* rseq, int off, int k, int capmask, bool insert) {
    const uint8_t* me = asm_ptr(ref, rseq, off);
    unsigned s = asm_hash(me, k) & (unsigned)capmask;
    for (;;) {
        int cur = S.key[s];
        if (cur == -1) {
            if (!insert) return -1;
            int old = atomicCAS(&S.key[s], -1, off);
            if (old == -1) return (int)s;
            cur = old;
        }
        if (cur == off || asm_eq(asm_ptr(ref, rseq, cur), me, k)) return (int)s;
        s = (s + 1u) & (unsigned)capmask;
    }
}

// does read r pass the k+1-base quality / N filter at position i (assembler.pyx:1362-1373)?  returns min qual or -1
__device__ __forceinline__ int asm_read_edge_q(const uint8_t* s, const uint8_t* q, int i, int k, int min_qual) {
    int mq = 100000000, hasn = 0;
    const int n = k + 1;
    for (int j = 0; j < n; j += 8) {
        unsigned long long wq = asm_ld8(q + i + j), ws = asm_ld8(s + i + j);
        const int m = n - j < 8 ? n - j : 8;
        for (int t = 0; t < m; ++t) {
            const int v = (int)(signed char)(wq & 0xFFu);
            mq = v < mq ? v : mq;
            hasn |= (ws & 0xFFu) == (unsigned long long)'N';
            wq >>= 8; ws >>= 8;
        }
    }
    return (mq >= min_qual && !hasn) ? mq : -1;
}

// ---- the LDS path works on an edge's k+1 bytes held in registers: KW 64-bit words, zero beyond the k+1 bytes (KW = 2 for k <= 15, the
// default: the only instantiation, a longer k takes the global path)
template <int KW> struct AsmWords { unsigned long long w[KW]; };
template <int KW> __device__ __forceinline__ AsmWords<KW> asm_load_words(const uint8_t* p, int nbytes) {
    AsmWords<KW> W;
#pragma unroll
    for (int c = 0; c < KW; ++c) W.w[c] = 8 * c < nbytes ? asm_ld8(p + 8 * c) & asm_tailmask(nbytes - 8 * c) : 0ull;
    return W;
}
// 8 bytes at byte offset o of the reference cached in LDS (aligned 64-bit reads + funnel)
__device__ __forceinline__ unsigned long long asm_lds8(const unsigned long long* s_ref, int o) {
    const int wi = o >> 3, sh = (o & 7) * 8;
    const unsigned long long a = s_ref[wi], b2 = s_ref[wi + 1];
    return sh ? (a >> sh) | (b2 << (64 - sh)) : a;
}
template <int KW> __device__ __forceinline__ AsmWords<KW> asm_load_words_lds(const unsigned long long* s_ref, int o, int nbytes) {
    AsmWords<KW> W;
#pragma unroll
    for (int c = 0; c < KW; ++c) W.w[c] = 8 * c < nbytes ? asm_lds8(s_ref, o + 8 * c) & asm_tailmask(nbytes - 8 * c) : 0ull;
    return W;
}
// The bytes [pos, pos + 8 KW) of a 256-byte window the wave holds one dword per lane (`d` = dword `lane` of the window): 2 KW + 1
// lane exchanges and a byte alignment instead of unaligned 64-bit loads per lane (which the texture addresser serialises).
template <int KW> __device__ __forceinline__ AsmWords<KW> asm_gather_words(unsigned d, int pos) {
    const int d0 = pos >> 2;
    const unsigned bo = (unsigned)pos & 3u;
    unsigned x[2 * KW + 1];
#pragma unroll
    for (int m = 0; m <= 2 * KW; ++m) x[m] = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (d0 + m), (int)d);
    AsmWords<KW> W;
#pragma unroll
    for (int c = 0; c < KW; ++c) {
        const unsigned lo = __builtin_amdgcn_alignbyte(x[2 * c + 1], x[2 * c], bo), hi = __builtin_amdgcn_alignbyte(x[2 * c + 2], x[2 * c + 1], bo);
        W.w[c] = (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    return W;
}
template <int KW> __device__ __forceinline__ AsmWords<KW> asm_mask_words(AsmWords<KW> W, int nbytes) {
#pragma unroll
    for (int c = 0; c < KW; ++c) W.w[c] = 8 * c < nbytes ? W.w[c] & asm_tailmask(nbytes - 8 * c) : 0ull;
    return W;
}
// the first k bytes of the words (the start k-mer), and bytes 1..k (the end k-mer)
template <int KW> __device__ __forceinline__ AsmWords<KW> asm_kmer_start(const AsmWords<KW>& E, int k) {
    AsmWords<KW> W;
#pragma unroll
    for (int c = 0; c < KW; ++c) W.w[c] = 8 * c < k ? E.w[c] & asm_tailmask(k - 8 * c) : 0ull;
    return W;
}
template <int KW> __device__ __forceinline__ AsmWords<KW> asm_kmer_end(const AsmWords<KW>& E, int k) {
    AsmWords<KW> W;
#pragma unroll
    for (int c = 0; c < KW; ++c) {
        const unsigned long long v = (E.w[c] >> 8) | (c + 1 < KW ? E.w[c + 1] << 56 : 0ull);
        W.w[c] = 8 * c < k ? v & asm_tailmask(k - 8 * c) : 0ull;
    }
    return W;
}
// Table hash of a k-mer: bits 1 and 2 of an ASCII base tell A, C, G, T apart (N falls on G, other bytes wherever: the comparison
// decides), so four dwords of bases pack into one word without losing anything; one multiplication scatters it.
template <int KW> __device__ __forceinline__ unsigned asm_hash_words(const AsmWords<KW>& K, int k) {
    unsigned h = 0u;
#pragma unroll
    for (int c = 0; c < KW; c += 2) {
        const unsigned x0 = (unsigned)K.w[c] & 0x06060606u, x1 = (unsigned)(K.w[c] >> 32) & 0x06060606u;
        const unsigned x2 = c + 1 < KW ? (unsigned)K.w[c + 1] & 0x06060606u : 0u, x3 = c + 1 < KW ? (unsigned)(K.w[c + 1] >> 32) & 0x06060606u : 0u;
        const unsigned p = (x0 >> 1) | (x1 << 1) | (x2 << 3) | (x3 << 5);
        h = (h << 7 | h >> 25) ^ p;
    }
    (void)k;
    h *= 0x9E3779B1u;
    return h ^ (h >> 15);
}
// does the k-mer K equal the k bytes at `off` (reference offset if isref, else read-blob offset)?
template <int KW> __device__ __forceinline__ bool asm_eq_words(const AsmWords<KW>& K, int k, bool isref, int off, const unsigned long long* s_ref, bool refc,
                                             const uint8_t* ref, const uint8_t* rseq) {
    unsigned long long diff = 0ull;
    if (isref && refc) {
#pragma unroll
        for (int c = 0; c < KW; ++c) if (8 * c < k) diff |= (asm_lds8(s_ref, off + 8 * c) & asm_tailmask(k - 8 * c)) ^ K.w[c];
    } else {
        const uint8_t* rp = isref ? ref + off : rseq + off;
#pragma unroll
        for (int c = 0; c < KW; ++c) if (8 * c < k) diff |= (asm_ld8(rp + 8 * c) & asm_tailmask(k - 8 * c)) ^ K.w[c];
    }
    return diff == 0ull;
}
// byte k of the words (the base an edge appends to its start k-mer)
template <int KW> __device__ __forceinline__ unsigned asm_byte_k(const AsmWords<KW>& E, int k) {
    unsigned long long w = E.w[0];
#pragma unroll
    for (int c = 1; c < KW; ++c) w = (k >> 3) == c ? E.w[c] : w;
    return (unsigned)(w >> (8 * (k & 7))) & 0xFFu;
}
// insert phase: tab[s] = byte offset of a representative occurrence (reads: 0x40000000 + blob offset), -1 empty; returns the slot
template <int KW> __device__ inline int asm_lds_insert_words(int* tab, const AsmWords<KW>& K, int k, int off, const unsigned long long* s_ref, bool refc,
                                            const uint8_t* ref, const uint8_t* rseq, int* distinct) {
    unsigned s = asm_hash_words(K, k) & (unsigned)(ASM_LDS_SLOTS - 1);
    for (;;) {
        int cur = tab[s];
        if (cur == -1) {
            const int old = atomicCAS(&tab[s], -1, off);
            if (old == -1) { atomicAdd(distinct, 1); return (int)s; }
            cur = old;
        }
        if (cur == off) return (int)s;
        const bool isref = cur < 0x40000000;
        if (asm_eq_words(K, k, isref, isref ? cur : cur - 0x40000000, s_ref, refc, ref, rseq)) return (int)s;
        s = (s + 1u) & (unsigned)(ASM_LDS_SLOTS - 1);
    }
}
// lookup phase: slots hold (node id << ASM_OFF_BITS | offset); ids below n_ref_nodes have their representative in the reference
template <int KW> __device__ inline int asm_lds_node_words(const int* tab, const AsmWords<KW>& K, int k, int n_ref_nodes, const unsigned long long* s_ref, bool refc,
                                         const uint8_t* ref, const uint8_t* rseq) {
    unsigned s = asm_hash_words(K, k) & (unsigned)(ASM_LDS_SLOTS - 1);
    for (;;) {
        const int v = tab[s];
        if (v == -1) return -1;
        const int id = (int)((unsigned)v >> ASM_OFF_BITS), o = v & ((1 << ASM_OFF_BITS) - 1);
        if (asm_eq_words(K, k, id < n_ref_nodes, o, s_ref, refc, ref, rseq)) return id;
        s = (s + 1u) & (unsigned)(ASM_LDS_SLOTS - 1);
    }
}
// Fused pass (round 4): the table holds its FINAL words from the start of the reads' pass -- node id << ASM_OFF_BITS | offset of the
// representative -- so that an edge's nodes are known the moment its k-mers are found: a k-mer met for the first time takes the next
// read-node id (an id whose insertion loses the race for the slot stays unused: a hole in the node arrays, nothing else).  Returns the
// slot, or -1 when the node arrays are full (the region is then redone on the global path).
template <int KW> __device__ inline int asm_lds_insert_final(int* tab, const AsmWords<KW>& K, int k, int roff, const unsigned long long* s_ref, bool refc,
                                            const uint8_t* ref, const uint8_t* rseq, int n_ref_nodes, int* n_read_nodes, int* rep) {
    unsigned s = asm_hash_words(K, k) & (unsigned)(ASM_LDS_SLOTS - 1);
    for (;;) {
        int v = tab[s];
        if (v == -1) {
            const int id = n_ref_nodes + atomicAdd(n_read_nodes, 1);
            if (id >= ASM_LDS_NODES) return -1;
            const int word = (int)(((unsigned)id << ASM_OFF_BITS) | (unsigned)roff);
            const int old = atomicCAS(&tab[s], -1, word);
            if (old == -1) { rep[id] = 0x40000000 + roff; return (int)s; }
            v = old;
        }
        const int id = (int)((unsigned)v >> ASM_OFF_BITS), o = v & ((1 << ASM_OFF_BITS) - 1);
        if (asm_eq_words(K, k, id < n_ref_nodes, o, s_ref, refc, ref, rseq)) return (int)s;
        s = (s + 1u) & (unsigned)(ASM_LDS_SLOTS - 1);
    }
}
// asm_read_edge_q on the words of the edge's k+1 bases (S) and qualities (Q): min quality, or -1 when the edge is filtered
template <int KW> __device__ __forceinline__ int asm_edge_q_words(const AsmWords<KW>& S, const AsmWords<KW>& Q, int k, int min_qual) {
    const int n = k + 1;
    unsigned mn = 255u;
    unsigned long long neg = 0ull, zero = 0ull;
#pragma unroll
    for (int c = 0; c < KW; ++c) {
        if (8 * c < n) {
            const unsigned long long m = asm_tailmask(n - 8 * c);
            const unsigned long long q = Q.w[c] | (0x7F7F7F7F7F7F7F7Full & ~m);          // bytes past the edge never lower the minimum
            neg |= q & 0x8080808080808080ull;
            const unsigned lo = (unsigned)q, hi = (unsigned)(q >> 32);
            mn = min(mn, min(min(lo & 0xFFu, (lo >> 8) & 0xFFu), min((lo >> 16) & 0xFFu, lo >> 24)));
            mn = min(mn, min(min(hi & 0xFFu, (hi >> 8) & 0xFFu), min((hi >> 16) & 0xFFu, hi >> 24)));
            const unsigned long long x = (S.w[c] ^ 0x4E4E4E4E4E4E4E4Eull) | ~m;            // a zero byte = an 'N' inside the edge
            zero |= (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
        }
    }
    if (neg) return -2;                                   // a quality byte >= 128 (negative as the reference reads it): the caller takes the byte loop
    return ((int)mn >= min_qual && !zero) ? (int)mn : -1;
}

// ---- quality minimum of every edge of a window at once (round 4) ------------------------------------------------------------------
// The window's quality bytes lie one dword per lane.  Instead of gathering each edge's k + 1 bytes and taking their minimum (a dozen lane
// exchanges and ~40 byte operations per edge), the lanes compute the SLIDING minimum over n = k + 1 bytes for all 256 positions together
// -- widths 2, 4, 8, 16 by doubling, each step one lane exchange and one byte-wise minimum --, and an edge fetches its byte.  Bytes must
// be < 128 (a quality >= 128 is negative as the reference reads it: such a window takes the per-edge code).
__device__ __forceinline__ unsigned asm_bytemin7(unsigned x, unsigned y) {            // per byte min(x, y), all bytes < 128
    const unsigned d = (x | 0x80808080u) - y;                                          // bit 7 of a byte: x >= y (no borrow crosses a byte)
    const unsigned m = (d >> 7) & 0x01010101u;
    const unsigned mask = (m << 8) - m;                                                // 0xFF where x >= y
    return (y & mask) | (x & ~mask);
}
// v's bytes [t, t + 4) of the window, for a dword-per-lane array (t in 0..11): the lanes that would read past lane 63 get their own value
__device__ __forceinline__ unsigned asm_window_shift(unsigned v, int t) {
    const unsigned a = (unsigned)__shfl_down((int)v, t >> 2), b2 = (unsigned)__shfl_down((int)v, (t >> 2) + 1);
    return __builtin_amdgcn_alignbyte(b2, a, (unsigned)(t & 3));
}
__device__ __forceinline__ unsigned asm_sliding_min(unsigned dq, int n) {            // n in 2..16: per byte position b, min of bytes [b, b + n)
    unsigned a = asm_bytemin7(dq, asm_window_shift(dq, 1));                            // width 2
    int wdt = 2;
    if (n >= 4) { a = asm_bytemin7(a, asm_window_shift(a, 2)); wdt = 4; }
    if (n >= 8) { a = asm_bytemin7(a, asm_window_shift(a, 4)); wdt = 8; }
    if (n >= 16) { a = asm_bytemin7(a, asm_window_shift(a, 8)); wdt = 16; }
    if (n > wdt) a = asm_bytemin7(a, asm_window_shift(a, n - wdt));                    // two overlapping windows of the largest width cover n
    return a;
}
// v's bytes [-t, -t + 4) of the window (t in 1..3): the mirror of asm_window_shift, for arrays that must move towards higher positions
__device__ __forceinline__ unsigned asm_window_shift_back(unsigned v, int t) {
    const unsigned prev = (unsigned)__shfl_up((int)v, 1);
    return __builtin_amdgcn_alignbyte(v, prev, (unsigned)(4 - t));
}
// position of the n-th (0-based) set bit of m; n < popcount(m)
__device__ __forceinline__ int asm_select64(unsigned long long m, int n) {
    unsigned wd = (unsigned)m;
    int pos = 0;
    const int c0 = __popc(wd);
    if (n >= c0) { n -= c0; wd = (unsigned)(m >> 32); pos = 32; }
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        const int c = __popc(wd & ((1u << sh) - 1u));
        if (n >= c) { n -= c; wd >>= sh; pos += sh; }
    }
    return pos;
}
// an 'N' among the edge's k + 1 bases?
template <int KW> __device__ __forceinline__ bool asm_edge_has_n(const AsmWords<KW>& S, int k) {
    const int n = k + 1;
    unsigned long long zero = 0ull;
#pragma unroll
    for (int c = 0; c < KW; ++c)
        if (8 * c < n) {
            const unsigned long long m = asm_tailmask(n - 8 * c);
            const unsigned long long x = (S.w[c] ^ 0x4E4E4E4E4E4E4E4Eull) | ~m;
            zero |= (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
        }
    return zero != 0ull;
}

// exclusive prefix sum of one value per thread over the workgroup (wave scans by lane shifts, the wave totals through `s_wsum`);
// every thread gets the total too.  All threads must call it.
__device__ __forceinline__ int asm_block_exscan(int v, int* s_wsum, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_wsum[wv] = x;
    __syncthreads();
    if (tid < 64) {
        int t = tid < nw ? s_wsum[tid] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(t, o); if (lane >= o) t += y; }
        s_wsum[tid] = t;
    }
    __syncthreads();
    const int before = wv ? s_wsum[wv - 1] : 0;
    total = s_wsum[nw - 1];
    __syncthreads();
    return before + x - v;
}

// workgroup barrier + agent-scope acquire: the graph lives in global memory and is updated with L2 atomics, so the
// CU's vector L1 must be invalidated before plain loads re-read it (MI355X_MICROARCH.md, inter-workgroup visibility;
// here producer and consumer are the same CU but the stale-L1 hazard is the same).
__device__ __forceinline__ void asm_sync() {
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// workgroup barrier alone (it carries the workgroup-scope release / acquire): for the phase boundaries where what the next phase reads
// from global memory was written with PLAIN stores by this workgroup (or is input) -- a CU's waves share its vector L1, which is
// write-through, so nothing can be stale.  The agent-scope invalidate above empties that L1 -- the next phase then fetches everything
// again from L2, its spilled registers included -- and with ~25 boundaries per region that was a quarter of the kernel's time: it stays
// only where words updated by L2 ATOMICS are read back by plain loads (the slot words before phase D; every boundary of the
// three-pass and global paths).
__device__ __forceinline__ void asm_sync_wg() { __syncthreads(); }

// measurement only (PLAT_ASM_TIMING=1): 100 MHz ticks the first thread of every workgroup spent up to each phase boundary, summed
__device__ unsigned long long g_asm_ticks[16];
// `tid` made opaque to the compiler (after every barrier): what it derived from the old value -- the 64-bit per-thread offsets of every
// strided loop -- is dead from there on instead of being carried, and spilled, across the kernel
#define ASM_FRESH() asm volatile("" : "+v"(tid))
#define ASM_TICK(i) do { ASM_FRESH(); if (P.timing && tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_asm_ticks[i], now_ - tick_); tick_ = now_; } } while (0)

__global__ void __launch_bounds__(ASM_THREADS)
k_assemble(plat_assembly_batch b, AsmParams P, char* scratch, int max_ref, int max_reads, int32_t* var_count,
           int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd, int32_t* var_off, uint8_t* var_blob,
           int32_t* status, const long long* __restrict__ verdict, long long* work, unsigned long long* wg_sig, unsigned long long sig)
{
    // (plat_assemble_batch_async: the sizes came from the caller instead of a read-back; k_asm_check left its verdict here -- a batch that
    //  does not fit them is refused as a whole, tile by tile, before anything is carved out of the scratch)
    if (verdict && verdict[3] != 0) {
        for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < b.n_regions; g += gridDim.x * blockDim.x) { status[g] = (int)verdict[3]; var_count[g] = 0; }
        return;
    }
    __shared__ int s_n, s_ntasks, s_err, s_cycle, s_k, s_pool, s_distinct, s_lds, s_nrefnodes, s_nreadnodes, s_nv;
    extern __shared__ __attribute__((aligned(16))) int s_tab[];  // [ASM_LDS_SLOTS] the k-mer table, then the node words, then the reference
    unsigned* s_first = (unsigned*)(s_tab + ASM_LDS_SLOTS);      // [ASM_LDS_NODES] min touch code (2*ticket + isEnd)
    unsigned* s_wc = s_first + ASM_LDS_NODES;                    // [ASM_LDS_NODES] weight | colour << 30
    unsigned long long* s_ref = (unsigned long long*)(s_wc + ASM_LDS_NODES);   // [ASM_REF_CACHE / 8] the region's reference bytes
    __shared__ int s_wsum[64];
    // (`tid` is made opaque at every phase boundary -- ASM_TICK -- so that what the compiler derives from it, 64-bit per-thread offsets of
    //  every strided loop, is recomputed per phase instead of being held, and spilled, across the whole kernel)
    int tid = threadIdx.x;
    const int nthr = blockDim.x;
    AsmScratch S = asm_carve(scratch + (size_t)blockIdx.x * P.scratch_per_block, P.cap, P.max_pos, max_ref, max_reads);
    const int capmask = P.cap - 1;
    unsigned long long tick_ = P.timing ? wall_clock64() : 0ull;
    // LDS path: the successor slot words of the first ASM_LDS_NODES nodes are kept CLEAN between regions (zeroed here once, and by
    // every region for the few nodes it dirtied): a node's first-claimed slot lives in LDS, so most nodes never touch theirs.
    // ... and between LAUNCHES: a workgroup that finished all its regions without an error leaves `sig` -- the layout of its slice of the
    // scratch (address, sizes, the allocation's epoch) -- in wg_sig[blockIdx.x]; the next launch that finds the same value there skips the
    // 1.4 MB of stores below (a third of the kernel's HBM traffic when a launch holds one tile per workgroup, as the region loop's do).
    __shared__ int s_dirty;
    const bool keep = wg_sig != nullptr && wg_sig[blockIdx.x] == sig && P.debug == 0 && P.fused;
    __syncthreads();
    if (tid == 0) { s_dirty = (P.debug != 0 || !P.fused) ? 1 : 0; if (wg_sig) wg_sig[blockIdx.x] = 0ull; }     // (nothing is promised while the launch runs)
    if (!keep)
    for (int i = tid; i < ASM_LDS_NODES * ASM_MAX_SUCC; i += nthr) { S.succ_cw[i] = 0ull; S.succ_c[i] = 0; S.succ_t[i] = 0xFFFFFFFFu; }
    // fused LDS path: the first ticket of a node's LDS-summed slot when a READ claimed it (read-only nodes; a reference-claimed slot's ticket
    // is the node's position) lives in S.weight[node] -- the LDS path has no other use for that array --, 0xFFFFFFFF between regions
    unsigned* own_t = (unsigned*)S.weight;
    // ... and the node that slot leads to in own_n[node] (S.succ_w, which only the global path sums into): one dense word per node
    // instead of a word in the node's row of eight -- a 32-byte sector written and read back per node
    int* own_n = S.succ_w;
    if (!keep) for (int i = tid; i < ASM_LDS_NODES; i += nthr) own_t[i] = 0xFFFFFFFFu;

    // Tiles are handed out as workgroups finish (one atomic per tile on a counter the launch zeroes): with 2 000 tiles for 256 workgroups a
    // static split gives some workgroups eight tiles and others seven, and tiles differ in depth and in what their bubbles hold.
    __shared__ int s_next_g;
    for (int g = blockIdx.x; g < b.n_regions;) {
        const uint8_t* ref = b.ref_seq + b.ref_off[g];
        const int refLen = (int)(b.ref_off[g + 1] - b.ref_off[g]);
        const int refStart = b.ref_start[g], aStart = b.assem_start[g], aEnd = b.assem_end[g];
        const int rb = b.reg_read_begin[g], nR = b.reg_read_begin[g + 1] - rb;
        const long long rblob0 = nR > 0 ? b.read_off[rb] : 0;
        const uint8_t* rseq = b.read_seq + rblob0;        // region-relative read blob
        const uint8_t* rqual = b.read_qual + rblob0;
        if (tid == 0) { s_err = 0; s_k = P.kmer; s_cycle = 0; }
        const bool refc = refLen + 24 <= ASM_REF_CACHE;   // the reference fits the LDS cache (with the slack 8-byte reads need)
        if (refc)
            for (int i = tid; 8 * i < refLen + 16; i += nthr) s_ref[i] = asm_ld8(ref + 8 * i);
        const long long blobLen = nR > 0 ? b.read_off[rb + nR] - rblob0 : 0;
        asm_sync_wg(); ASM_FRESH();

        // Fast path of the phases after the graph is built (round 4; fused LDS path, noCycles off): the k-mer table is not needed once the
        // successors are picked, so its 64 KB take (a) ONE WORD PER NODE -- end node | weight >= min_weight << 14 | number of out-edges << 15
        // -- which is the whole out-edge list of the nodes with at most one successor (all but the branch points, whose AsmNodeE stays in
        // global memory), (b) the tasks' finished-path counts and (c) the first ASM_LDS_ARENA path elements of the bubble walks.  First
        // touches and colours are read from the node words in LDS.  The walks of phases E-G are chains of dependent loads: in LDS a step
        // costs ~0.1 us instead of a global round trip.
        // Layout of the table's words on the fast path: [0, nNodes) edge words, then one finished-path count per task, then the tasks' stacks
        // of pending path elements (when they fit: else in the slice), then path elements (node, parent, depth) up to the table's end.
        bool fast = false, ldsStk = false, ldsPath = false;
        bool relax = false;                                  // the fused LDS path is running: phase boundaries without the L1 invalidate (asm_sync_wg)
        auto sync_phase = [&]() { if (relax) asm_sync_wg(); else asm_sync(); };
        int* const s_edge = s_tab;
        int* s_tnfin = s_tab; int* s_stk = s_tab; int* s_path = s_tab; int* s_arena = s_tab;
        int arenaCap = 0;
        auto colour_of = [&](int n) -> int { return fast ? (int)(s_wc[n] >> 30) : S.colour[n]; };
        auto first_of = [&](int n) -> unsigned { return fast ? s_first[n] : S.first[n]; };
        auto edges_of = [&](int n, int (&end)[4], bool (&heavy)[4]) -> int {            // out-edges in pick order; heavy: weight >= min_weight
            if (fast) {
                const int wd = s_edge[n], cnt = wd >> 15 & 7;
                if (cnt <= 1) { end[0] = wd & 0x3FFF; heavy[0] = (wd >> 14 & 1) != 0; return cnt; }
            }
            const AsmNodeE& E = S.edges[n];
            const int cnt = E.n;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i < cnt) { end[i] = E.end[i]; heavy[i] = E.w[i] >= P.min_weight; }
            return cnt;
        };
        auto arena_get = [&](int e, int f) -> int { return e < arenaCap ? s_arena[3 * e + f] : S.arena[3 * e + f]; };
        auto arena_set = [&](int e, int node, int parent, int depth) {
            int* A = e < arenaCap ? s_arena + 3 * e : S.arena + 3 * e;
            A[0] = node; A[1] = parent; A[2] = depth;
        };

        for (;;) {   // (re)build with the current k (assembler.pyx:1453-1469: k += 5 while cycles, noCycles only)
            const int k = s_k;
            const int nRefE = refLen - k - 1 > 0 ? refLen - k - 1 : 0;
            // ticket bases of the reads: exclusive scan of max(L - k - 1, 0) (each thread sums a run of reads, thread 0 scans the
            // per-thread sums in LDS, each thread writes its run)
            {
                const int per = (nR + nthr - 1) / nthr;
                const int r0 = min(nR, tid * per), r1 = min(nR, r0 + per);
                int acc = 0;
                for (int r = r0; r < r1; ++r) {
                    const int L = (int)(b.read_off[rb + r + 1] - b.read_off[rb + r]);
                    acc += L - k - 1 > 0 ? L - k - 1 : 0;
                }
                int total;
                acc = asm_block_exscan(acc, s_wsum, total);
                if (tid == 0) { S.read_base[nR] = total; s_n = 0; s_ntasks = 0; s_pool = 0; }
                for (int r = r0; r < r1; ++r) {
                    S.read_base[r] = acc;
                    const int L = (int)(b.read_off[rb + r + 1] - b.read_off[rb + r]);
                    acc += L - k - 1 > 0 ? L - k - 1 : 0;
                }
            }
            asm_sync_wg(); ASM_FRESH();
            ASM_TICK(0);
            const int nReadE = S.read_base[nR];
            const int nEv = nRefE + nReadE;
            if ((long long)nEv + 2ll * (nR + 1) > (long long)P.max_pos) {                       // distinct k-mers <= occurrences
                if (tid == 0) s_err = PLAT_ERR_OVERFLOW;
                asm_sync(); ASM_FRESH();
                break;
            }
            // every AddEdge event: the reference's edges one per thread, then the reads one per wave at a time (lanes over the read's
            // edges: no search for the read of an event).  fn(ticket, start offset, end offset, colour, weight) -> false stops this thread.
            auto for_each_event = [&](auto&& fn) {
                for (int e = tid; e < nRefE; e += nthr)
                    if (!fn(e, e, e + 1, 1, 1)) return;
                const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
                for (int r = wv; r < nR; r += nwv) {
                    const int base = S.read_base[r], cnt = S.read_base[r + 1] - base;
                    const int ro = (int)(b.read_off[rb + r] - rblob0);
                    for (int i = lane; i < cnt; i += 64) {
                        const int w = asm_read_edge_q(rseq + ro, rqual + ro, i, k, P.min_qual);
                        if (w < 0) continue;
                        if (!fn(nRefE + base + i, 0x40000000 + ro + i, 0x40000000 + ro + i + 1, 2, w)) return;
                    }
                }
            };
            // LDS path: the edges of the reads, one read per wave at a time.  A window of 256 bytes of the read's bases and one of its
            // qualities are loaded one aligned dword per lane; the lanes then take the window's edges in rounds of 64, each gathering
            // its k+1 bytes from the other lanes.  stage1(i, ro, E, w) -> tag >= 0 for every valid edge; stage2(ticket offset, ro + i,
            // E, w, tag, tag of the next edge or -1) after the wave has exchanged tags; skipped(ticket offset) for a filtered edge.
            auto read_pass = [&](auto KWc, auto&& stage1, auto&& stage2, auto&& skipped, auto&& stop) {
                constexpr int KW = decltype(KWc)::value;
                const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
                constexpr int WIN = 4 * (64 - 2 * KW) - 3;     // edges per window: the last one still finds its 2 KW + 1 dwords in lanes <= 63
                // A wave's reads are a chain of dependent round trips (the read's offsets, then its bytes): the offsets of the read after
                // next and the first window of the next read are requested before the current read is worked on.
                struct Meta { int base, cnt, ro; };
                struct Win { unsigned dS, dQ; int sS, sQ, nE; };
                auto load_meta = [&](int r) -> Meta {
                    Meta m{0, 0, 0};
                    if (r < nR) { m.base = S.read_base[r]; m.cnt = S.read_base[r + 1] - m.base; m.ro = (int)(b.read_off[rb + r] - rblob0); }
                    return m;
                };
                auto load_win = [&](const Meta& m, int c0) -> Win {
                    Win w;
                    w.nE = max(0, min(WIN, m.cnt - c0));
                    const uintptr_t pS = (uintptr_t)(rseq + m.ro + c0), pQ = (uintptr_t)(rqual + m.ro + c0);
                    w.sS = (int)(pS & 3); w.sQ = (int)(pQ & 3);
                    // (lanes past the window's last needed byte load nothing: the blobs' slack is a few bytes, not a window)
                    w.dS = (w.nE > 0 && 4 * lane < w.sS + w.nE + k + 1) ? *(const unsigned*)((pS & ~(uintptr_t)3) + 4 * lane) : 0u;
                    w.dQ = (w.nE > 0 && 4 * lane < w.sQ + w.nE + k + 1) ? *(const unsigned*)((pQ & ~(uintptr_t)3) + 4 * lane) : 0u;
                    return w;
                };
                Meta m0 = load_meta(wv), m1 = load_meta(wv + nwv);
                Win w0 = load_win(m0, 0);
                for (int r = wv; r < nR; r += nwv) {
                    const Meta m2 = load_meta(r + 2 * nwv);
                    const Win w1 = load_win(m1, 0);
                    const int base = m0.base, cnt = m0.cnt, ro = m0.ro;
                    for (int c0 = 0; c0 < cnt; c0 += WIN) {
                        if (stop()) return;
                        const Win w = c0 == 0 ? w0 : load_win(m0, c0);
                        const int nE = w.nE;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (64 * u < nE) {                              // wave-uniform
                                if (u > 0 && stop()) return;                // (checked every round: between two checks the workgroup inserts
                                                                            //  at most 2 x 1024 k-mers, which the table's spare slots take)
                                const int j = 64 * u + lane, i = c0 + j;
                                const AsmWords<KW> E = asm_mask_words(asm_gather_words<KW>(w.dS, j + w.sS), k + 1);
                                const AsmWords<KW> Q = asm_mask_words(asm_gather_words<KW>(w.dQ, j + w.sQ), k + 1);
                                int wq = -1, tag = -1;
                                if (j < nE) {
                                    wq = asm_edge_q_words(E, Q, k, P.min_qual);
                                    if (wq == -2) wq = asm_read_edge_q(rseq + ro, rqual + ro, i, k, P.min_qual);
                                    if (wq >= 0) tag = stage1(i, ro, E, wq);
                                }
                                int ntag = __shfl_down(tag, 1);
                                if (lane == 63) ntag = -1;
                                if (wq >= 0) stage2(base + i, ro + i, E, wq, tag, ntag);
                                else if (j < nE) skipped(base + i);
                            }
                        }
                    }
                    m0 = m1; m1 = m2; w0 = w1;
                }
            };
            // ---- phase A: insert every k-mer that takes part in a (valid) edge; LDS table first, the global one if it overflows
            // (or if k, the reference or the reads' bytes are beyond what the LDS path packs into its words)
            if (tid == 0) s_lds = (k <= 15 && nRefE < ASM_LDS_LIMIT && refLen < (1 << ASM_OFF_BITS) && blobLen < (1ll << ASM_OFF_BITS)) ? 1 : 0;
            __syncthreads();
            // ---- LDS path, fused (round 4).  Rounds 2-3 inserted every k-mer (phase A), numbered the nodes (B), and only then ran the AddEdge
            // events (C) from one 4-byte word per edge that phase A had left in global memory -- 0.5 MB per region written, read, and read
            // again for the first tickets (D): two thirds of the kernel's HBM traffic.  Now:
            //   1. the reference's k-mers are inserted and its nodes numbered along the reference (bitmap + ranks, as before); the table then
            //      holds final words;
            //   2. the reference's events: first touches, barrier -- every reference node's position is final, reads only come later --, then
            //      the successor slots: the edge that leaves a node at its FIRST reference occurrence claims the node's LDS-summed slot (bit
            //      27 of its word); its first ticket is the node's position and needs no word of its own.  Edges of later occurrences (repeats)
            //      take the global slot words, with their first ticket (atomicMin);
            //   3. the reads, ONE pass: an edge finds or creates its two nodes (asm_lds_insert_final) and is applied at once.  First tickets are
            //      kept (one global atomicMin) for every slot but a reference-claimed LDS slot, whose ticket no read can lower;
            //   4. phase D picks the successors from those tickets: no pass over the events, no event words for the reads at all.
            bool fused_done = false;
            relax = false;
            if (P.fused && s_lds) {
                relax = true;
                for (int i = tid; i < ASM_LDS_SLOTS; i += nthr) s_tab[i] = -1;
                for (int i = tid; i < ASM_LDS_NODES; i += nthr) { s_first[i] = 0xFFFFFFFFu; s_wc[i] = 0u; }
                if (tid == 0) { s_distinct = 0; s_nrefnodes = 0; s_nreadnodes = 0; }
                asm_sync_wg(); ASM_FRESH();
                int* ev = S.stack;                        // one word per REFERENCE edge (slot of its start k-mer | 1 << 14 | base << 22 | end flag << 30)
                int* ev_end = S.stack + P.max_pos;
                constexpr int KW = 2;
                for (int e = tid; e < nRefE; e += nthr) {
                    const AsmWords<KW> E = refc ? asm_load_words_lds<KW>(s_ref, e, k + 1) : asm_load_words<KW>(ref + e, k + 1);
                    const int slot = asm_lds_insert_words(s_tab, asm_kmer_start(E, k), k, e, s_ref, refc, ref, rseq, &s_distinct);
                    int word = slot | 1 << 14 | (int)(asm_byte_k(E, k) & 0xFFu) << 22;
                    if (e == nRefE - 1) {
                        ev_end[e] = asm_lds_insert_words(s_tab, asm_kmer_end(E, k), k, e + 1, s_ref, refc, ref, rseq, &s_distinct);
                        word |= 1 << 30;
                    }
                    ev[e] = word;
                }
                asm_sync_wg(); ASM_FRESH();
                // the reference's nodes in the order of their representatives along the reference (ids 0 .. nRefNodes0 - 1)
                const int refBytes = ((refLen + 16 + 7) >> 3) << 3, nW32 = (refLen >> 5) + 1;
                const bool ordered = refc && refBytes + 8 * nW32 <= ASM_REF_CACHE && nW32 <= nthr;
                unsigned* s_bm = (unsigned*)((char*)s_ref + refBytes);
                unsigned* s_bp = s_bm + nW32;
                if (ordered) for (int i = tid; i < nW32; i += nthr) s_bm[i] = 0u;
                __syncthreads();
                int nRefNodes0 = 0;
                if (ordered) {
                    for (int sidx = tid; sidx < ASM_LDS_SLOTS; sidx += nthr) {
                        const int off = s_tab[sidx];
                        if (off != -1) atomicOr(&s_bm[off >> 5], 1u << (off & 31));
                    }
                    __syncthreads();
                    const int mine = tid < nW32 ? __popc(s_bm[tid]) : 0;
                    const int before = asm_block_exscan(mine, s_wsum, nRefNodes0);
                    if (tid < nW32) s_bp[tid] = (unsigned)before;
                    __syncthreads();
                } else nRefNodes0 = s_distinct;
                for (int sidx = tid; sidx < ASM_LDS_SLOTS; sidx += nthr) {
                    const int off = s_tab[sidx];
                    if (off != -1) {
                        const int id = ordered ? (int)s_bp[off >> 5] + __popc(s_bm[off >> 5] & ((1u << (off & 31)) - 1u)) : atomicAdd(&s_nrefnodes, 1);
                        S.rep[id] = off;
                        s_tab[sidx] = (int)(((unsigned)id << ASM_OFF_BITS) | (unsigned)off);
                    }
                }
                __syncthreads();
                if (tid == 0) s_nrefnodes = nRefNodes0;
                auto node_of = [&](int slot) -> int { return (int)((unsigned)s_tab[slot] >> ASM_OFF_BITS); };
                // slot of a successor byte; bytes other than A, C, G, T share slots 4..7 (claimed in the node's global byte word)
                auto succ_slot = [&](int sn, unsigned c) -> int {
                    if (c == 'A' || c == 'C' || c == 'G' || c == 'T') return (int)((c >> 1) & 3u);
                    unsigned* cw = (unsigned*)(S.succ_c + (size_t)sn * ASM_MAX_SUCC) + 1;            // bytes 4..7
                    for (int j = 0; j < 4; ++j)
                        for (;;) {
                            const unsigned wd = *(volatile unsigned*)cw;
                            const unsigned cur = (wd >> (8 * j)) & 0xFFu;
                            if (cur == c) return 4 + j;
                            if (cur != 0u) break;
                            if (atomicCAS(cw, wd, wd | (c << (8 * j))) == wd) return 4 + j;
                        }
                    return -1;
                };
                // (the first event of a slot also notes where the slot leads: phase D then neither hashes nor compares)
                auto global_slot = [&](int sn, int slot, int w, int e, int en) {
                    if (!(s_wc[sn] >> 26 & 1u)) atomicOr(&s_wc[sn], 1u << 26);                       // (bit 26: the node's global slot words are in use)
                    // (every event of a slot stores the same end node: a plain store -- asking the atomic for its old value to store it once
                    //  would make the wave wait a global round trip whenever one of its 64 edges leaves the main path, i.e. most rounds)
                    atomicAdd(&S.succ_cw[sn * ASM_MAX_SUCC + slot], (1ull << 32) | (unsigned long long)(unsigned)w);
                    S.succ_n[sn * ASM_MAX_SUCC + slot] = en;
                    atomicMin(&S.succ_t[sn * ASM_MAX_SUCC + slot], (unsigned)e);
                };
                // 2a. first touches and colours of the reference's events
                for (int e = tid; e < nRefE; e += nthr) {
                    const int word = ev[e];
                    const int sn = node_of(word & 0x3FFF);
                    const int en = node_of((word >> 30 & 1) ? ev_end[e] : (ev[e + 1] & 0x3FFF));
                    atomicMin(&s_first[sn], 2u * (unsigned)e);
                    atomicMin(&s_first[en], 2u * (unsigned)e + 1u);
                    if ((s_wc[sn] >> 30 & 1u) == 0u) atomicOr(&s_wc[sn], 1u << 30);
                    if ((s_wc[en] >> 30 & 1u) == 0u) atomicOr(&s_wc[en], 1u << 30);
                    S.ref_node[e] = sn;
                    if (e == nRefE - 1) S.ref_node[e + 1] = en;
                }
                __syncthreads();
                // 2b. successor slots of the reference's events
                for (int e = tid; e < nRefE; e += nthr) {
                    const int word = ev[e];
                    const int sn = node_of(word & 0x3FFF);
                    const int en = node_of((word >> 30 & 1) ? ev_end[e] : (ev[e + 1] & 0x3FFF));
                    const unsigned c = (unsigned)(word >> 22) & 0xFFu;
                    const int slot = succ_slot(sn, c);
                    if (slot < 0) { s_err = PLAT_ERR_UNSUPPORTED; continue; }
                    if (slot >= 4 && !(s_wc[sn] >> 26 & 1u)) atomicOr(&s_wc[sn], 1u << 26);          // its byte word is in use
                    const unsigned ft = s_first[sn];
                    if (slot < 7 && (int)(ft >> 1) + (int)(ft & 1u) == e) {                           // the node's first occurrence: the one claim of its LDS slot in this pass
                        atomicAdd(&s_wc[sn], ((unsigned)(slot + 1) << 23) | (1u << 27) | 1u);
                        own_n[sn] = en;
                    } else global_slot(sn, slot, 1, e, en);
                }
                asm_sync_wg(); ASM_FRESH();
                ASM_TICK(1);
                // 3. the reads
                bool bad = false;
                {
                    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
                    // A wave takes one read at a time; a window of 256 bytes of its bases and of its qualities is held one aligned dword per
                    // lane (as in read_pass).  The window's edges that pass the quality / N rule are compacted and worked NRB rounds of 64 side by
                    // side (`work` below): the pass is bound by vector issue plus chains of LDS round trips (gather -> table -> representative ->
                    // node words), and independent chains per lane share each wait (four rounds at once spilled registers: two are kept).
                    constexpr int WIN = 4 * (64 - 2 * KW) - 3;
                    struct Meta { int base, cnt, ro; };
                    struct Win { unsigned dS, dQ; int sS, sQ, nE; };
                    auto load_meta = [&](int r) -> Meta {
                        Meta m{0, 0, 0};
                        if (r < nR) { m.base = S.read_base[r]; m.cnt = S.read_base[r + 1] - m.base; m.ro = (int)(b.read_off[rb + r] - rblob0); }
                        return m;
                    };
                    auto load_win = [&](const Meta& m, int c0) -> Win {
                        Win w;
                        w.nE = max(0, min(WIN, m.cnt - c0));
                        const uintptr_t pS = (uintptr_t)(rseq + m.ro + c0), pQ = (uintptr_t)(rqual + m.ro + c0);
                        w.sS = (int)(pS & 3); w.sQ = (int)(pQ & 3);
                        w.dS = (w.nE > 0 && 4 * lane < w.sS + w.nE + k + 1) ? *(const unsigned*)((pS & ~(uintptr_t)3) + 4 * lane) : 0u;
                        w.dQ = (w.nE > 0 && 4 * lane < w.sQ + w.nE + k + 1) ? *(const unsigned*)((pQ & ~(uintptr_t)3) + 4 * lane) : 0u;
                        return w;
                    };
                    // (two windows ahead instead of one was measured: no faster -- the loads are not late -- and eight more live registers)
                    Meta m0 = load_meta(wv), m1 = load_meta(wv + nwv);
                    Win w0 = load_win(m0, 0);
                    bool stopped = false;
#ifdef PLAT_ASM_SECTIONS                                                     // (measurement builds: shader-clock cycles of the first wave per section of the loop;
                    unsigned long long sec_[5] = {0, 0, 0, 0, 0}, secT_ = clock64();         //  compiled out by default: twelve live registers in a kernel that spills)
#define ASM_SEC(i) do { const unsigned long long n_ = clock64(); sec_[i] += n_ - secT_; secT_ = n_; } while (0)
#else
#define ASM_SEC(i) do { } while (0)
#endif
                    for (int r = wv; r < nR && !stopped; r += nwv) {
                        const Meta m2 = load_meta(r + 2 * nwv);
                        const Win w1 = load_win(m1, 0);
                        const int base = m0.base, cnt = m0.cnt, ro = m0.ro;
                        for (int c0 = 0; c0 < cnt; c0 += WIN) {
                            if (nRefNodes0 + *(volatile int*)&s_nreadnodes > ASM_LDS_LIMIT) { stopped = true; break; }   // (<= 4 x 1024 new k-mers between two looks: the arrays' spare room)
                            const Win w = c0 == 0 ? w0 : load_win(m0, c0);
                            const int nE = w.nE;
                            constexpr int NRB = ASM_THREADS <= 512 ? 4 : 2;      // edges per lane worked side by side (registers: 254 per lane at 512 threads, 128 at 1024)
                            // `work`: NRB edges per lane (jj[u] = edge index in the window, -1 none; ww[u] = its weight): k-mers found or created,
                            // then the events.  An edge's end k-mer is the start k-mer of the edge that follows it in the read -- the next lane's
                            // (or lane 0's of the next u) when that lane holds edge jj + 1.
                            auto work = [&](const int (&jj)[NRB], const int (&ww)[NRB]) {
                                ASM_SEC(1);
                                AsmWords<KW> E[NRB];
                                int slots[NRB];
#pragma unroll
                                for (int u = 0; u < NRB; ++u) {
                                    slots[u] = -1;
                                    E[u] = asm_mask_words(asm_gather_words<KW>(w.dS, (jj[u] >= 0 ? jj[u] : 0) + w.sS), k + 1);
                                }
                                ASM_SEC(2);
                                // the start k-mers, found or created: NRB probe sequences per lane side by side
                                if (!(P.debug & 2)) {
                                    unsigned sl[NRB]; bool todo[NRB];
                                    AsmWords<KW> Ks[NRB];
#pragma unroll
                                    for (int u = 0; u < NRB; ++u) { Ks[u] = asm_kmer_start(E[u], k); todo[u] = jj[u] >= 0; sl[u] = asm_hash_words(Ks[u], k) & (unsigned)(ASM_LDS_SLOTS - 1); }
                                    for (;;) {
                                        bool anytodo = false;
#pragma unroll
                                        for (int u = 0; u < NRB; ++u) anytodo |= todo[u];
                                        if (!anytodo) break;
                                        int v[NRB];
#pragma unroll
                                        for (int u = 0; u < NRB; ++u) v[u] = todo[u] ? s_tab[sl[u]] : 0;
#pragma unroll
                                        for (int u = 0; u < NRB; ++u)
                                            if (todo[u] && v[u] == -1) {                           // an empty slot: a k-mer met for the first time
                                                const int id = nRefNodes0 + atomicAdd(&s_nreadnodes, 1);
                                                if (id >= ASM_LDS_NODES) { todo[u] = false; continue; }   // (slots[u] stays -1: the region is redone on the global path)
                                                const int roff = ro + c0 + jj[u];
                                                const int old = atomicCAS(&s_tab[sl[u]], -1, (int)(((unsigned)id << ASM_OFF_BITS) | (unsigned)roff));
                                                if (old == -1) { S.rep[id] = 0x40000000 + roff; slots[u] = (int)sl[u]; todo[u] = false; }
                                                else v[u] = old;
                                            }
                                        bool eq[NRB];
#pragma unroll
                                        for (int u = 0; u < NRB; ++u) {
                                            const int id = (int)((unsigned)v[u] >> ASM_OFF_BITS), o = v[u] & ((1 << ASM_OFF_BITS) - 1);
                                            eq[u] = todo[u] && asm_eq_words(Ks[u], k, id < nRefNodes0, o, s_ref, refc, ref, rseq);
                                        }
#pragma unroll
                                        for (int u = 0; u < NRB; ++u)
                                            if (todo[u]) {
                                                if (eq[u]) { slots[u] = (int)sl[u]; todo[u] = false; }
                                                else sl[u] = (sl[u] + 1u) & (unsigned)(ASM_LDS_SLOTS - 1);
                                            }
                                    }
                                }
                                ASM_SEC(3);
                                int nsl[NRB];
#pragma unroll
                                for (int u = 0; u < NRB; ++u) {
                                    int nx = __shfl_down(slots[u], 1), jn = __shfl_down(jj[u], 1);
                                    if (lane == 63) {
                                        nx = u + 1 < NRB ? __shfl(slots[u + 1 < NRB ? u + 1 : u], 0) : -1;
                                        jn = u + 1 < NRB ? __shfl(jj[u + 1 < NRB ? u + 1 : u], 0) : -1;
                                    }
                                    nsl[u] = (jn == jj[u] + 1) ? nx : -1;
                                }
#pragma unroll
                                for (int u = 0; u < NRB; ++u) {
                                    if (jj[u] < 0 || slots[u] < 0 || (P.debug & 1)) continue;
                                    const int off = ro + c0 + jj[u];
                                    int nslot = nsl[u];
                                    if (nslot < 0) nslot = asm_lds_insert_final(s_tab, asm_kmer_end(E[u], k), k, off + 1, s_ref, refc, ref, rseq, nRefNodes0, &s_nreadnodes, S.rep);
                                    if (nslot < 0) continue;                                           // node arrays full: the region is redone below
                                    const int w_ = ww[u];
                                    const int e = nRefE + base + c0 + jj[u], sn = node_of(slots[u]), en = node_of(nslot);
                                    atomicMin(&s_first[sn], 2u * (unsigned)e);
                                    atomicMin(&s_first[en], 2u * (unsigned)e + 1u);
                                    unsigned x = s_wc[sn];
                                    atomicOr(&s_wc[sn], 2u << 30);                                  // (unconditionally: an LDS atomic without a return costs less than
                                    atomicOr(&s_wc[en], 2u << 30);                                  //  reading the word back and branching on it: +2.6 %)
                                    const unsigned c = asm_byte_k(E[u], k) & 0xFFu;
                                    const int slot = succ_slot(sn, c);
                                    if (slot < 0) { bad = true; continue; }                            // > 4 distinct other bytes
                                    if (slot >= 4 && !(x >> 26 & 1u)) atomicOr(&s_wc[sn], 1u << 26);
                                    bool local = false;
                                    if (slot < 7) {
                                        for (;;) {
                                            const unsigned d = (x >> 23) & 7u;
                                            if (d == 0u) {
                                                const unsigned old = atomicCAS(&s_wc[sn], x, x | (unsigned)(slot + 1) << 23);
                                                if (old == x) own_n[sn] = en;                              // this event claimed the slot
                                                x = old == x ? (x | (unsigned)(slot + 1) << 23) : old;
                                                continue;
                                            }
                                            local = d == (unsigned)slot + 1u && (x & 0x7FFFFFu) < 0x780000u;
                                            break;
                                        }
                                    }
                                    if (local) {
                                        atomicAdd(&s_wc[sn], (unsigned)w_);
                                        if (!(x >> 27 & 1u)) {                                         // a read claimed this slot: its first ticket in the node's own word (bit 28: in use)
                                            if (!(x >> 28 & 1u)) atomicOr(&s_wc[sn], 1u << 28);
                                            atomicMin(&own_t[sn], (unsigned)e);
                                        }
                                    } else global_slot(sn, slot, w_, e, en);
                                }
                                ASM_SEC(4);
                            };
                            // The quality / N filter (assembler.pyx:1362-1373) for ALL edges of the window at once: qualities with the N positions
                            // zeroed, sliding minimum over k + 1 bytes (needs every quality byte < 128 and min_qual >= 1: else edge by edge below).
                            // With 5 % of the bases below Q20 more than half of the edges are filtered: the edges that pass are then COMPACTED --
                            // lane l of batch cb takes the (64 cb + l)-th passing edge (n-th set bit of the rounds' ballots) -- so that the probes
                            // and events run with full waves (measured before: a third of the lanes active per vector instruction).
                            const bool q7 = k >= 1 && P.min_qual >= 1 && !__any((w.dQ & 0x80808080u) != 0u);
                            if (q7) {
                                const int delta = w.sS - w.sQ;                                      // bases in the qualities' byte positions
                                const unsigned bq = delta == 0 ? w.dS : (delta > 0 ? asm_window_shift(w.dS, delta) : asm_window_shift_back(w.dS, -delta));
                                const unsigned xn = bq ^ 0x4E4E4E4Eu;                                 // a zero byte = an 'N'
                                const unsigned nz = ~(((xn & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xn | 0x7F7F7F7Fu);   // 0x80 in exactly the zero bytes
                                const unsigned nmask = ((nz >> 7) << 8) - (nz >> 7);                  // 0xFF per N byte
                                const unsigned qmin4 = asm_sliding_min(w.dQ & ~nmask, k + 1);
                                unsigned long long vm[4];
                                int mqv[4], cum[5];
                                cum[0] = 0;
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const int j = 64 * u + lane, pq = j + w.sQ;
                                    const unsigned dwq = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (pq >> 2), (int)qmin4);
                                    mqv[u] = (int)((dwq >> (8 * (pq & 3))) & 0xFFu);
                                    vm[u] = __ballot(j < nE && mqv[u] >= P.min_qual);
                                    cum[u + 1] = cum[u] + __popcll(vm[u]);
                                }
                                const int V = cum[4];
                                const unsigned mq4 = (unsigned)mqv[0] | (unsigned)mqv[1] << 8 | (unsigned)mqv[2] << 16 | (unsigned)mqv[3] << 24;   // (each < 128)
                                for (int cb = 0; cb < V; cb += 64 * NRB) {
                                    int jj[NRB], ww[NRB];
#pragma unroll
                                    for (int u = 0; u < NRB; ++u) {
                                        const int c = cb + 64 * u + lane;
                                        jj[u] = -1; ww[u] = 0;
                                        // round of the c-th passing edge, its lane there, and its weight from that lane
                                        const int ru = c < cum[1] ? 0 : (c < cum[2] ? 1 : (c < cum[3] ? 2 : 3));
                                        const unsigned long long mk = ru == 0 ? vm[0] : (ru == 1 ? vm[1] : (ru == 2 ? vm[2] : vm[3]));
                                        const int before = ru == 0 ? cum[0] : (ru == 1 ? cum[1] : (ru == 2 ? cum[2] : cum[3]));
                                        const int src = c < V ? asm_select64(mk, c - before) : 0;
                                        const unsigned got = (unsigned)__builtin_amdgcn_ds_bpermute(4 * src, (int)mq4);   // (the four rounds' weights of lane src, a byte each)
                                        if (c < V) { jj[u] = 64 * ru + src; ww[u] = (int)((got >> (8 * ru)) & 0xFFu); }
                                    }
                                    work(jj, ww);
                                }
                            } else {
                                for (int ub = 0; ub < 4 && 64 * ub < nE; ub += NRB) {
                                    int jj[NRB], ww[NRB];
#pragma unroll
                                    for (int u = 0; u < NRB; ++u) {
                                        const int j = 64 * (ub + u) + lane, i = c0 + j;
                                        jj[u] = -1; ww[u] = 0;
                                        const AsmWords<KW> Eq = asm_mask_words(asm_gather_words<KW>(w.dS, j + w.sS), k + 1);
                                        const AsmWords<KW> Q = asm_mask_words(asm_gather_words<KW>(w.dQ, j + w.sQ), k + 1);
                                        if (j < nE) {
                                            int wq = asm_edge_q_words(Eq, Q, k, P.min_qual);
                                            if (wq == -2) wq = asm_read_edge_q(rseq + ro, rqual + ro, i, k, P.min_qual);
                                            if (wq >= 0) { jj[u] = j; ww[u] = wq; }
                                        }
                                    }
                                    work(jj, ww);
                                }
                            }
                        }
                        m0 = m1; m1 = m2; w0 = w1;
                        ASM_SEC(0);
                    }
#ifdef PLAT_ASM_SECTIONS
                    if (P.timing && tid == 0) for (int i = 0; i < 5; ++i) atomicAdd(&g_asm_ticks[11 + i], sec_[i]);
#endif
#undef ASM_SEC
                }
                if (bad) s_err = PLAT_ERR_UNSUPPORTED;
                asm_sync_wg(); ASM_FRESH();
                if (nRefNodes0 + s_nreadnodes > ASM_LDS_LIMIT) {
                    // more nodes than the LDS takes: leave the global slot words clean and take the global path
                    for (int n = tid; n < ASM_LDS_NODES; n += nthr) {
                        if (s_wc[n] >> 28 & 1u) own_t[n] = 0xFFFFFFFFu;
                        if (s_wc[n] >> 26 & 1u)
                            for (int j = 0; j < ASM_MAX_SUCC; ++j) { S.succ_cw[n * ASM_MAX_SUCC + j] = 0ull; S.succ_c[n * ASM_MAX_SUCC + j] = 0; S.succ_t[n * ASM_MAX_SUCC + j] = 0xFFFFFFFFu; }
                    }
                    __syncthreads();
                    if (tid == 0) s_lds = 0;
                    relax = false;
                } else {
                    if (tid == 0) s_n = nRefNodes0 + s_nreadnodes;
                    fused_done = true;
                }
                asm_sync(); ASM_FRESH();
                ASM_TICK(2);
            }
            bool failed = false;
            for (;;) {
                if (fused_done) break;
                if (tid == 0) s_dirty = 1;                   // (the three-pass and global paths tidy up too, but only the fused path's word is given across launches)
                const bool lds = s_lds != 0;
                if (lds) { for (int i = tid; i < ASM_LDS_SLOTS; i += nthr) s_tab[i] = -1; if (tid == 0) s_distinct = 0; }
                else {
                    if ((long long)nEv * 4 > (long long)P.cap * 3) { failed = true; break; }
                    for (int i = tid; i < P.cap; i += nthr) S.key[i] = -1;
                }
                asm_sync(); ASM_FRESH();
                if (lds) {
                    // every edge leaves a word for phase C: table slot of its start k-mer | weight << 14 | appended base << 22 |
                    // (its end k-mer was inserted by this edge and its slot is in ev_end) << 30;  -1 for a filtered edge
                    int* ev = S.stack;
                    int* ev_end = S.stack + P.max_pos;
                    auto insert_all = [&](auto KWc) {
                        constexpr int KW = decltype(KWc)::value;
                        // the reference's k-mers first: a k-mer the reference holds is then represented by a reference occurrence, and
                        // comparing a read's k-mer with its representative reads the LDS copy of the reference
                        for (int e = tid; e < nRefE; e += nthr) {
                            const AsmWords<KW> E = refc ? asm_load_words_lds<KW>(s_ref, e, k + 1) : asm_load_words<KW>(ref + e, k + 1);
                            const int slot = asm_lds_insert_words(s_tab, asm_kmer_start(E, k), k, e, s_ref, refc, ref, rseq, &s_distinct);
                            int word = slot | 1 << 14 | (int)(asm_byte_k(E, k) & 0xFFu) << 22;
                            if (e == nRefE - 1) {
                                ev_end[e] = asm_lds_insert_words(s_tab, asm_kmer_end(E, k), k, e + 1, s_ref, refc, ref, rseq, &s_distinct);
                                word |= 1 << 30;
                            }
                            ev[e] = word;
                        }
                        __syncthreads();
                        // the reads: the end k-mer of an edge is the start k-mer of the next one, which inserts it unless it is filtered
                        read_pass(KWc,
                            [&](int i, int ro, const AsmWords<KW>& E, int w) -> int {
                                (void)w;
                                return asm_lds_insert_words(s_tab, asm_kmer_start(E, k), k, 0x40000000 + ro + i, s_ref, refc, ref, rseq, &s_distinct);
                            },
                            [&](int t, int off, const AsmWords<KW>& E, int w, int slot, int nslot) {
                                int word = slot | (w & 0xFF) << 14 | (int)(asm_byte_k(E, k) & 0xFFu) << 22;
                                if (nslot < 0) {
                                    ev_end[nRefE + t] = asm_lds_insert_words(s_tab, asm_kmer_end(E, k), k, 0x40000000 + off + 1, s_ref, refc, ref, rseq, &s_distinct);
                                    word |= 1 << 30;
                                }
                                ev[nRefE + t] = word;
                            },
                            [&](int t) { ev[nRefE + t] = -1; },
                            [&]() -> bool { return *(volatile int*)&s_distinct > ASM_LDS_LIMIT; });
                    };
                    insert_all(std::integral_constant<int, 2>{});
                } else {
                    for_each_event([&](int e, int so, int eo, int col, int w) -> bool {
                        (void)w;
                        asm_slot(S, ref, rseq, so, k, capmask, true);
                        if (col == 2 || e == nRefE - 1) asm_slot(S, ref, rseq, eo, k, capmask, true);
                        return true;
                    });
                }
                asm_sync(); ASM_FRESH();
                if (!lds || s_distinct <= ASM_LDS_LIMIT) break;
                __syncthreads();
                if (tid == 0) s_lds = 0;                                    // more distinct k-mers than the LDS table takes: global table
                __syncthreads();
            }
            if (failed) {
                if (tid == 0) s_err = PLAT_ERR_OVERFLOW;
                asm_sync(); ASM_FRESH();
                break;
            }
            ASM_TICK(1);
            const bool lds = s_lds != 0;
            // ---- phase B: dense node ids + field initialisation
            auto init_node = [&](int id, int rep_off) {
                S.rep[id] = rep_off;
                if (lds) return;                          // (first touch / colours come from the LDS words; the slot words are clean)
                S.first[id] = 0xFFFFFFFFu; S.weight[id] = 0; S.colour[id] = 0;
                for (int j = 0; j < ASM_MAX_SUCC; ++j) {
                    S.succ_c[id * ASM_MAX_SUCC + j] = 0; S.succ_t[id * ASM_MAX_SUCC + j] = 0xFFFFFFFFu;
                    S.succ_w[id * ASM_MAX_SUCC + j] = 0; S.succ_n[id * ASM_MAX_SUCC + j] = -1;
                }
            };
            if (fused_done) {
                // (the fused pass numbered the nodes as it went)
            } else if (lds) {
                // Nodes represented by a reference occurrence get the low ids (a slot then tells where its representative lives), in the
                // order of their representatives along the reference when the LDS has room for a bitmap of the positions: reads arrive
                // sorted by position and walk along the reference, so the successor slots they update then lie next to each other in
                // memory and stay in the L2 (with ids in table order every update was an L2 miss).
                const int refBytes = ((refLen + 16 + 7) >> 3) << 3, nW32 = (refLen >> 5) + 1;
                const bool ordered = refc && refBytes + 8 * nW32 <= ASM_REF_CACHE && nW32 <= nthr;
                unsigned* s_bm = (unsigned*)((char*)s_ref + refBytes);       // [nW32] bit p: reference position p represents a node
                unsigned* s_bp = s_bm + nW32;                                // [nW32] nodes before the word
                if (tid == 0) { s_nrefnodes = 0; s_nreadnodes = 0; }
                if (ordered) for (int i = tid; i < nW32; i += nthr) s_bm[i] = 0u;
                __syncthreads();
                int nRefNodes0;
                if (ordered) {
                    for (int sidx = tid; sidx < ASM_LDS_SLOTS; sidx += nthr) {
                        const int off = s_tab[sidx];
                        if (off != -1 && off < 0x40000000) atomicOr(&s_bm[off >> 5], 1u << (off & 31));
                    }
                    __syncthreads();
                    const int mine = tid < nW32 ? __popc(s_bm[tid]) : 0;
                    const int before = asm_block_exscan(mine, s_wsum, nRefNodes0);
                    if (tid < nW32) s_bp[tid] = (unsigned)before;
                    __syncthreads();
                } else {
                    int mine = 0;
                    for (int sidx = tid; sidx < ASM_LDS_SLOTS; sidx += nthr) { const int off = s_tab[sidx]; mine += off != -1 && off < 0x40000000; }
                    if (mine) atomicAdd(&s_nrefnodes, mine);
                    __syncthreads();
                    nRefNodes0 = s_nrefnodes;
                    __syncthreads();
                    if (tid == 0) s_nrefnodes = 0;
                    __syncthreads();
                }
                for (int sidx = tid; sidx < ASM_LDS_SLOTS; sidx += nthr) {
                    const int off = s_tab[sidx];
                    if (off != -1) {
                        const bool isref = off < 0x40000000;
                        int id;
                        if (!isref) id = nRefNodes0 + atomicAdd(&s_nreadnodes, 1);
                        else if (ordered) id = (int)s_bp[off >> 5] + __popc(s_bm[off >> 5] & ((1u << (off & 31)) - 1u));
                        else id = atomicAdd(&s_nrefnodes, 1);
                        init_node(id, off);
                        s_tab[sidx] = (int)(((unsigned)id << ASM_OFF_BITS) | (unsigned)(isref ? off : off - 0x40000000));
                        s_first[id] = 0xFFFFFFFFu; s_wc[id] = 0u;
                    }
                }
                __syncthreads();
                if (tid == 0) { s_nrefnodes = nRefNodes0; s_n = nRefNodes0 + s_nreadnodes; }
            } else {
                for (int sidx = tid; sidx < P.cap; sidx += nthr)
                    if (S.key[sidx] != -1) { const int id = atomicAdd(&s_n, 1); S.slot_id[sidx] = id; init_node(id, S.key[sidx]); }
            }
            if (fused_done) asm_sync_wg(); else asm_sync();       // (fused: only s_n, in LDS, was written since the fence that ended the pass over the reads)
            ASM_FRESH();
            ASM_TICK(2);
            const int nNodes = s_n;
            // ---- phase C: AddEdge events (assembler.pyx:801-827)
            const int nRefNodes = s_nrefnodes;
            if (fused_done) {
                // (... and applied the events)
            } else if (lds) {
                // one event per thread, from the word phase A left: node words in LDS; the successor slot of the start node is picked by the
                // byte the edge appends (A, C, G, T: slots 0..3 by the byte's bits; anything else shares slots 4..7 by search), and takes ONE
                // global atomic: its event count and weight.  Which node a slot leads to is looked up once per slot in phase D; first
                // tickets only matter where a node has several successors and are collected for those nodes alone in a second pass.
                const int* ev = S.stack;
                const int* ev_end = S.stack + P.max_pos;
                // (four tickets per thread and trip, their words loaded before the first is used)
                for (int e0 = tid; e0 < nEv; e0 += 4 * nthr) {
                  int words[4], nexts[4];
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                      const int e = e0 + u * nthr;
                      words[u] = e < nEv ? ev[e] : -1;
                      nexts[u] = e + 1 < nEv ? ev[e + 1] : -1;
                  }
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * nthr, word = words[u];
                    if (word < 0) continue;
                    const int w = (word >> 14) & 0xFF, col = e < nRefE ? 1 : 2;
                    const unsigned c = (unsigned)(word >> 22) & 0xFFu;
                    const int sn = (int)((unsigned)s_tab[word & 0x3FFF] >> ASM_OFF_BITS);
                    const int en = (int)((unsigned)s_tab[(word >> 30 & 1) ? ev_end[e] : (nexts[u] & 0x3FFF)] >> ASM_OFF_BITS);
                    atomicMin(&s_first[sn], 2u * (unsigned)e);
                    atomicMin(&s_first[en], 2u * (unsigned)e + 1u);
                    // (a node's own weight, assembler.pyx:795, is never read again: only its colours are kept)
                    unsigned x = s_wc[sn];
                    if ((x >> 30 & (unsigned)col) == 0u) atomicOr(&s_wc[sn], (unsigned)col << 30);
                    if ((s_wc[en] >> 30 & (unsigned)col) == 0u) atomicOr(&s_wc[en], (unsigned)col << 30);
                    if (e < nRefE) { S.ref_node[e] = sn; if (e == nRefE - 1) S.ref_node[e + 1] = en; }
                    int slot = -1;
                    if (c == 'A' || c == 'C' || c == 'G' || c == 'T') slot = (int)((c >> 1) & 3u);
                    else {
                        unsigned* cw = (unsigned*)(S.succ_c + (size_t)sn * ASM_MAX_SUCC) + 1;         // bytes 4..7
                        for (int j = 0; j < 4 && slot < 0; ++j) {
                            for (;;) {
                                const unsigned wd = *(volatile unsigned*)cw;
                                const unsigned cur = (wd >> (8 * j)) & 0xFFu;
                                if (cur == c) { slot = 4 + j; break; }
                                if (cur != 0u) break;
                                const unsigned old = atomicCAS(cw, wd, wd | (c << (8 * j)));
                                if (old == wd) { slot = 4 + j; break; }
                            }
                        }
                        if (slot < 0) { s_err = PLAT_ERR_UNSUPPORTED; continue; }                     // > 4 distinct other bytes
                        if (!(x >> 26 & 1u)) atomicOr(&s_wc[sn], 1u << 26);                           // the node's global slot words are in use
                    }
                    // The weight of the node's FIRST-CLAIMED successor slot is summed in the node's LDS word (bits 0..22, the slot + 1 in
                    // bits 23..25): the edges along the path most reads take never leave the CU.  Other slots, and a sum about to leave
                    // its 23 bits, take a global atomic on the slot's (event count, weight) word.
                    bool local = false;
                    if (slot < 7) {
                        for (;;) {
                            const unsigned d = (x >> 23) & 7u;
                            if (d == 0u) {
                                const unsigned old = atomicCAS(&s_wc[sn], x, x | (unsigned)(slot + 1) << 23);
                                x = old == x ? (x | (unsigned)(slot + 1) << 23) : old;
                                continue;
                            }
                            local = d == (unsigned)slot + 1u && (x & 0x7FFFFFu) < 0x780000u;
                            break;
                        }
                    }
                    if (local) atomicAdd(&s_wc[sn], (unsigned)w);
                    else {
                        if (!(x >> 26 & 1u)) atomicOr(&s_wc[sn], 1u << 26);                           // (bit 26: global slot words in use)
                        atomicAdd(&S.succ_cw[sn * ASM_MAX_SUCC + slot], (1ull << 32) | (unsigned long long)(unsigned)w);
                    }
                  }
                }
            } else
            for_each_event([&](int e, int so, int eo, int col, int w) -> bool {
                const int sn = S.slot_id[asm_slot(S, ref, rseq, so, k, capmask, false)];
                const int en = S.slot_id[asm_slot(S, ref, rseq, eo, k, capmask, false)];
                atomicMin(&S.first[sn], 2u * (unsigned)e);
                atomicMin(&S.first[en], 2u * (unsigned)e + 1u);
                atomicAdd(&S.weight[sn], w); atomicAdd(&S.weight[en], w);
                atomicOr(&S.colour[sn], col); atomicOr(&S.colour[en], col);
                if (e < nRefE) { S.ref_node[e] = sn; if (e == nRefE - 1) S.ref_node[e + 1] = en; }
                // successor slot keyed by the byte appended to the start k-mer
                const unsigned char c = asm_ptr(ref, rseq, eo)[k - 1];
                unsigned* cw = (unsigned*)(S.succ_c + (size_t)sn * ASM_MAX_SUCC);       // 8 bytes = 2 dwords
                int slot = -1;
                for (int j = 0; j < ASM_MAX_SUCC && slot < 0; ++j) {
                    for (;;) {
                        const unsigned word = cw[j >> 2];
                        const unsigned cur = (word >> (8 * (j & 3))) & 0xFFu;
                        if (cur == c) { slot = j; break; }
                        if (cur != 0u) break;
                        const unsigned old = atomicCAS(&cw[j >> 2], word, word | ((unsigned)c << (8 * (j & 3))));
                        if (old == word) { slot = j; break; }
                    }
                }
                if (slot < 0) { s_err = PLAT_ERR_UNSUPPORTED; return true; }             // > 8 distinct successor bytes
                // (a stale look at the ticket only costs the atomic it could have saved)
                if (S.succ_t[sn * ASM_MAX_SUCC + slot] > (unsigned)e) atomicMin(&S.succ_t[sn * ASM_MAX_SUCC + slot], (unsigned)e);
                atomicAdd(&S.succ_w[sn * ASM_MAX_SUCC + slot], w);
                S.succ_n[sn * ASM_MAX_SUCC + slot] = en;
                return true;
            });
            if (!fused_done) { asm_sync(); ASM_FRESH(); }        // (fused: phase C is empty)
            ASM_TICK(3);
            fast = lds && fused_done && !P.no_cycles;
            if (lds && !fast) {                                               // the node words the later phases read, to the slice
                for (int n = tid; n < nNodes; n += nthr) { S.first[n] = s_first[n]; if (!fused_done) S.weight[n] = 0; S.colour[n] = (int)(s_wc[n] >> 30); }
                asm_sync(); ASM_FRESH();
            }
            ASM_TICK(9);
            // ---- phase D: per node, the four successors with the smallest first tickets, in ticket order
            if (lds) {
                // nodes with more than one successor slot in use (few): their slots' first tickets from a second pass over the events
                // (the fused path counts a node's used slots where it picks them, from the slot words it has loaded anyway: no pass here)
                for (int n = tid; n < nNodes && !fused_done; n += nthr) {
                    if (!(s_wc[n] >> 26 & 1u)) continue;               // only its LDS slot, or none
                    int used = 0;
                    const int own = (int)(s_wc[n] >> 23 & 7u) - 1;      // the slot summed in the LDS word, -1 none
                    for (int j = 0; j < ASM_MAX_SUCC; ++j) used += j == own || (S.succ_cw[n * ASM_MAX_SUCC + j] >> 32) != 0ull;
                    if (used > 1) {
                        s_wc[n] |= 1u << 29;
                        if (!fused_done) for (int j = 0; j < ASM_MAX_SUCC; ++j) S.succ_t[n * ASM_MAX_SUCC + j] = 0xFFFFFFFFu;
                    }
                }
                if (!fused_done) { sync_phase(); ASM_FRESH(); }
                ASM_TICK(10);
                if (!fused_done) {                                  // (the fused pass kept the first tickets as it went)
                    const int* ev = S.stack;
                    for (int e0 = tid; e0 < nEv; e0 += 4 * nthr) {
                      int words[4];
#pragma unroll
                      for (int u = 0; u < 4; ++u) words[u] = e0 + u * nthr < nEv ? ev[e0 + u * nthr] : -1;
#pragma unroll
                      for (int u = 0; u < 4; ++u) {
                        const int e = e0 + u * nthr, word = words[u];
                        if (word < 0) continue;
                        const int sn = (int)((unsigned)s_tab[word & 0x3FFF] >> ASM_OFF_BITS);
                        if (!(s_wc[sn] >> 29 & 1u)) continue;
                        const unsigned c = (unsigned)(word >> 22) & 0xFFu;
                        int slot = -1;
                        if (c == 'A' || c == 'C' || c == 'G' || c == 'T') slot = (int)((c >> 1) & 3u);
                        else for (int j = 4; j < ASM_MAX_SUCC; ++j) if (S.succ_c[sn * ASM_MAX_SUCC + j] == c) slot = j;
                        if (slot >= 0) atomicMin(&S.succ_t[sn * ASM_MAX_SUCC + slot], (unsigned)e);
                      }
                    }
                }
                if (!fused_done) { asm_sync(); ASM_FRESH(); }           // (fused: nothing was written since the fence that ended the pass over the reads)
                auto pick_edges = [&](auto KWc) {
                    constexpr int KW = decltype(KWc)::value;
                    for (int n = tid; n < nNodes; n += nthr) {
                        AsmNodeE E; E.n = 0;
                        const bool several = (s_wc[n] >> 29 & 1u) != 0u, dirty = (s_wc[n] >> 26 & 1u) != 0u;
                        const int own = (int)(s_wc[n] >> 23 & 7u) - 1;
                        int rep_off = -1;
                        AsmWords<KW> R;
                        unsigned last = 0; bool firstpick = true;
                        for (int pick = 0; pick < 4; ++pick) {
                            int bj = -1; unsigned bt = 0xFFFFFFFFu;
                            for (int j = 0; j < ASM_MAX_SUCC; ++j) {
                                if (j != own && (!dirty || (S.succ_cw[n * ASM_MAX_SUCC + j] >> 32) == 0ull)) continue;
                                // first ticket of the slot; fused pass: a node's reference-claimed LDS slot (bit 27) is the edge that leaves
                                // its first reference occurrence, whose ticket is the node's position
                                unsigned t = 0u;
                                if (several) {
                                    if (fused_done && j == own && (s_wc[n] >> 27 & 1u)) { const unsigned ft = s_first[n]; t = (ft >> 1) + (ft & 1u); }
                                    else t = S.succ_t[n * ASM_MAX_SUCC + j];
                                }
                                if (!firstpick && t <= last) continue;
                                if (t < bt) { bt = t; bj = j; }
                            }
                            if (bj < 0) break;
                            // the node the slot leads to: this node's k-mer without its first base, plus the slot's byte
                            if (rep_off < 0) {
                                rep_off = S.rep[n];
                                const bool isref = rep_off < 0x40000000;
                                R = (isref && refc) ? asm_load_words_lds<KW>(s_ref, rep_off, k + 1)
                                                    : asm_load_words<KW>(isref ? ref + rep_off : rseq + (rep_off - 0x40000000), k + 1);
                            }
                            const unsigned c = bj < 4 ? (unsigned)"ACTG"[bj] : (unsigned)S.succ_c[n * ASM_MAX_SUCC + bj];
                            AsmWords<KW> T = asm_kmer_end(R, k);
#pragma unroll
                            for (int cc = 0; cc < KW; ++cc)
                                if (((k - 1) >> 3) == cc) T.w[cc] = (T.w[cc] & ~(0xFFull << (8 * ((k - 1) & 7)))) | ((unsigned long long)c << (8 * ((k - 1) & 7)));
                            E.end[E.n] = asm_lds_node_words(s_tab, T, k, nRefNodes, s_ref, refc, ref, rseq);
                            E.w[E.n] = (dirty ? (int)(unsigned)S.succ_cw[n * ASM_MAX_SUCC + bj] : 0) + (bj == own ? (int)(s_wc[n] & 0x7FFFFFu) : 0);
                            ++E.n;
                            last = bt; firstpick = false;
                            if (!several) break;
                        }
                        S.edges[n] = E;
                        if (dirty)                                  // leave the node's slot words clean for the next region
                            for (int j = 0; j < ASM_MAX_SUCC; ++j) { S.succ_cw[n * ASM_MAX_SUCC + j] = 0ull; S.succ_c[n * ASM_MAX_SUCC + j] = 0; S.succ_t[n * ASM_MAX_SUCC + j] = 0xFFFFFFFFu; }
                    }
                };
                if (!fused_done) pick_edges(std::integral_constant<int, 2>{});
                else {
                    // fused pass: every used slot knows its end node (succ_n).  Eight nodes per thread and trip, the one global word most of them
                    // need -- the end of their only slot -- requested for all eight before the first is used
                    constexpr int DB = 8;
                    for (int n0 = tid; n0 < nNodes; n0 += DB * nthr) {
                        unsigned xs[DB]; int ends[DB];
#pragma unroll
                        for (int u = 0; u < DB; ++u) {
                            const int n = n0 + u * nthr;
                            xs[u] = n < nNodes ? s_wc[n] : 0u;
                            const int own = (int)(xs[u] >> 23 & 7u) - 1;
                            ends[u] = (n < nNodes && own >= 0) ? own_n[n] : -1;
                        }
#pragma unroll
                        for (int u = 0; u < DB; ++u) {
                            const int n = n0 + u * nthr;
                            if (n >= nNodes) continue;
                            const unsigned x = xs[u];
                            const bool dirty = (x >> 26 & 1u) != 0u;
                            bool several = false;                                                  // more than one slot in use (counted below)
                            const int own = (int)(x >> 23 & 7u) - 1;
                            // (written field by field: a local AsmNodeE indexed by the pick count would live in scratch memory)
                            AsmNodeE* Eo = &S.edges[n];
                            int en_ = 0, end0 = 0, w0 = 0;
                            if (!dirty) {
                                if (own >= 0) {
                                    end0 = ends[u]; w0 = (int)(x & 0x7FFFFFu); en_ = 1;
                                    if (!fast) { Eo->end[0] = end0; Eo->w[0] = w0; }           // (fast path: the node's edge word says it all)
                                }
                                if (!fast) Eo->n = en_;
                            } else {
                                // the node's slot words, all requested before the first is looked at (a branch point's picks were a chain of
                                // dependent global round trips: most of this phase's time)
                                unsigned long long cw[ASM_MAX_SUCC]; unsigned tt[ASM_MAX_SUCC]; int nn[ASM_MAX_SUCC];
#pragma unroll
                                for (int j = 0; j < ASM_MAX_SUCC; ++j) {
                                    cw[j] = S.succ_cw[n * ASM_MAX_SUCC + j]; tt[j] = S.succ_t[n * ASM_MAX_SUCC + j]; nn[j] = S.succ_n[n * ASM_MAX_SUCC + j];
                                }
                                {
                                    int used = 0;
#pragma unroll
                                    for (int j = 0; j < ASM_MAX_SUCC; ++j) used += j == own || (cw[j] >> 32) != 0ull;
                                    several = used > 1;
                                }
                                unsigned last = 0; bool firstpick = true;
                                for (int pick = 0; pick < 4; ++pick) {
                                    int bj = -1, bend = 0, bw = 0; unsigned bt = 0xFFFFFFFFu;
#pragma unroll
                                    for (int j = 0; j < ASM_MAX_SUCC; ++j) {
                                        if (j != own && (cw[j] >> 32) == 0ull) continue;
                                        unsigned t = 0u;
                                        if (several) {
                                            t = tt[j];                                                      // (events of the slot that went to the global words)
                                            if (j == own) {
                                                if (x >> 27 & 1u) { const unsigned ft = s_first[n]; t = (ft >> 1) + (ft & 1u); }
                                                else t = min(t, own_t[n]);
                                            }
                                        }
                                        if (!firstpick && t <= last) continue;
                                        if (t < bt) { bt = t; bj = j; bend = j == own ? ends[u] : nn[j]; bw = (int)(unsigned)cw[j] + (j == own ? (int)(x & 0x7FFFFFu) : 0); }
                                    }
                                    if (bj < 0) break;
                                    Eo->end[en_] = bend;
                                    Eo->w[en_] = bw;
                                    if (en_ == 0) { end0 = bend; w0 = bw; }
                                    ++en_;
                                    last = bt; firstpick = false;
                                    if (!several) break;
                                }
                                Eo->n = en_;
#pragma unroll
                                for (int j = 0; j < ASM_MAX_SUCC; ++j) { S.succ_cw[n * ASM_MAX_SUCC + j] = 0ull; S.succ_c[n * ASM_MAX_SUCC + j] = 0; S.succ_t[n * ASM_MAX_SUCC + j] = 0xFFFFFFFFu; }
                            }
                            if (fast) s_edge[n] = en_ > 0 ? (end0 | (w0 >= P.min_weight ? 1 << 14 : 0) | en_ << 15) : 0;
                            if (x >> 28 & 1u) own_t[n] = 0xFFFFFFFFu;
                        }
                    }
                }
            } else
            for (int n = tid; n < nNodes; n += nthr) {
                AsmNodeE E; E.n = 0;
                unsigned last = 0; bool firstpick = true;
                for (int pick = 0; pick < 4; ++pick) {
                    int bj = -1; unsigned bt = 0xFFFFFFFFu;
                    for (int j = 0; j < ASM_MAX_SUCC; ++j) {
                        const unsigned t = S.succ_t[n * ASM_MAX_SUCC + j];
                        if (t == 0xFFFFFFFFu) continue;
                        if (!firstpick && t <= last) continue;
                        if (t < bt) { bt = t; bj = j; }
                    }
                    if (bj < 0) break;
                    E.end[E.n] = S.succ_n[n * ASM_MAX_SUCC + bj]; E.w[E.n] = S.succ_w[n * ASM_MAX_SUCC + bj]; ++E.n;
                    last = bt; firstpick = false;
                }
                S.edges[n] = E;
                if (n < ASM_LDS_NODES) S.weight[n] = -1;            // (= 0xFFFFFFFF: the fused LDS path's per-node ticket word)
                if (n < ASM_LDS_NODES)                              // the LDS path of the regions that follow expects these words clean
                    for (int j = 0; j < ASM_MAX_SUCC; ++j) { S.succ_cw[n * ASM_MAX_SUCC + j] = 0ull; S.succ_c[n * ASM_MAX_SUCC + j] = 0; S.succ_t[n * ASM_MAX_SUCC + j] = 0xFFFFFFFFu; }
            }
            sync_phase(); ASM_FRESH();
            ASM_TICK(4);
            // ---- noCycles: detectCyclesInGraph_Recursive (assembler.pyx:831-898), iterative, one thread
            if (P.no_cycles) {
                if (tid == 0) {
                    int cyc = 0;
                    for (int n = 0; n < nNodes; ++n) S.dfs[n] = 'w';
                    for (int n0 = 0; n0 < nNodes && !cyc; ++n0) {
                        if (S.dfs[n0] != 'w') continue;
                        int sp = 0;
                        S.stack[0] = n0; S.stack[1] = 0; sp = 1; S.dfs[n0] = 'g';
                        while (sp > 0 && !cyc) {
                            const int n = S.stack[2 * (sp - 1)];
                            int ei = S.stack[2 * (sp - 1) + 1];
                            const AsmNodeE& E = S.edges[n];
                            bool descended = false;
                            while (ei < E.n) {
                                const int m = E.end[ei]; const int wgt = E.w[ei]; ++ei;
                                if (S.colour[m] == 2 && wgt < P.min_weight) continue;
                                if (S.dfs[m] == 'w') {
                                    S.stack[2 * (sp - 1) + 1] = ei;
                                    S.stack[2 * sp] = m; S.stack[2 * sp + 1] = 0; ++sp; S.dfs[m] = 'g';
                                    descended = true; break;
                                } else if (S.dfs[m] == 'g') { cyc = 1; break; }
                            }
                            if (!descended && !cyc) { S.dfs[n] = 'b'; --sp; }
                        }
                    }
                    s_cycle = cyc;
                    if (cyc && k <= 50) s_k = k + 5;
                }
                asm_sync(); ASM_FRESH();
                if (s_cycle && k <= 50) continue;          // rebuild with a longer k
                if (s_cycle) break;                        // k > 50 and still cyclic: no variants (assembler.pyx:1454-1457)
            }
            ASM_TICK(5);
            // ---- phase E: bubble starts in allNodes order (= increasing position for REF_AND_READ nodes): positions in chunks of
            // nthr, one thread per position, the tasks of a chunk written in position order behind those of the chunk before
            {
                const int i0 = aStart - refStart > 0 ? aStart - refStart : 0;
                const int i1 = aEnd - refStart < nRefE + 1 ? aEnd - refStart : nRefE + 1;
                for (int c0 = i0; c0 < i1 && nRefE > 0; c0 += nthr) {
                    const int i = c0 + tid;
                    int n = -1, cnt = 0, mask = 0;
                    if (i < i1) {
                        n = S.ref_node[i];
                        const unsigned ft = first_of(n);
                        if ((int)(ft >> 1) + (int)(ft & 1u) == i && colour_of(n) == 3) {       // first occurrence of this k-mer; assembler.pyx:1144
                            int end[4]; bool heavy[4];
                            const int ne = edges_of(n, end, heavy);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < ne && colour_of(end[j]) == 2) { mask |= 1 << j; ++cnt; }   // assembler.pyx:1153
                        }
                    }
                    int total;
                    const int sofar = s_ntasks;
                    int at = sofar + asm_block_exscan(cnt, s_wsum, total);
                    if (sofar + total > ASM_MAX_TASKS) { at = ASM_MAX_TASKS; if (tid == 0) s_err = PLAT_ERR_OVERFLOW; }
                    else if (tid == 0) s_ntasks = sofar + total;
                    for (int j = 0; j < 4 && cnt > 0; ++j)
                        if ((mask >> j) & 1) { if (at < ASM_MAX_TASKS) { S.task_node[at] = n; S.task_edge[at] = j; } ++at; }
                    __syncthreads();
                }
            }
            sync_phase(); ASM_FRESH();
            ASM_TICK(6);
            const int nTasks = s_ntasks;
            arenaCap = 0; ldsStk = false; ldsPath = false;
            if (fast) {
                s_tnfin = s_tab + nNodes;
                s_stk = s_tnfin + nTasks;
                ldsStk = nNodes + nTasks + ASM_STK * nTasks + 3 * 256 <= ASM_LDS_SLOTS;
                if (P.debug & 4) ldsStk = false;                                      // (tests: the stacks in the slice, no path arrays ...
                // ... and (when they fit too) the walks' current paths, 16 bytes aligned for the vector loads of the cycle check
                s_path = s_tab + ((nNodes + nTasks + ASM_STK * nTasks + 3) & ~3);
                ldsPath = ldsStk && (int)(s_path - s_tab) + (ASM_PATH / 2) * nTasks + 3 * 256 <= ASM_LDS_SLOTS;
                s_arena = ldsPath ? s_path + (ASM_PATH / 2) * nTasks : (ldsStk ? s_stk + ASM_STK * nTasks : s_stk);
                arenaCap = (int)((s_tab + ASM_LDS_SLOTS - s_arena) / 3);
                if (P.debug & 4) arenaCap = arenaCap < 24 ? arenaCap : 24;           //  ... and all but a few path elements in the global arena)
            }
            // ---- phase F: getVariantPathsThroughGraphFromNode (assembler.pyx:1027-1112), one thread per start edge
            for (int t = tid; t < nTasks; t += nthr) {
                // (the stack of pending elements in LDS, or in the slice: a local array indexed by `top` would live in scratch memory,
                //  a global round trip for every push and pop)
                int* const stk = ldsStk ? s_stk + ASM_STK * t : S.stack + ASM_STK * t;
                // The nodes of the CURRENT path by depth (pn[d], d = 1 .. ASM_PATH - 1).  The walk pops last-pushed-first, so between
                // the expansion of an element of depth d - 1 and the pop of one of its children only elements of depth >= d are popped:
                // when an element of depth d is popped, pn[1 .. d - 1] are its ancestors.  checkPathForCycles is then a scan of that
                // array -- loads that do not depend on each other -- instead of a chain of parent links, one LDS round trip each.
                unsigned short* const pn = ldsPath ? (unsigned short*)(s_path + (ASM_PATH / 2) * t) : nullptr;
                int top = 0, nfin = 0;
                const int e0 = atomicAdd(&s_pool, 2);
                bool aborted = false, overflow = e0 + 2 > ASM_POOL;
                if (!overflow) {
                    const int tn = S.task_node[t], te = S.task_edge[t];
                    int end[4]; bool heavy[4];
                    edges_of(tn, end, heavy);
                    const int first = te == 0 ? end[0] : (te == 1 ? end[1] : (te == 2 ? end[2] : end[3]));
                    arena_set(e0, tn, -1, 1);
                    arena_set(e0 + 1, first, e0, 2);
                    stk[top++] = e0 + 1;
                    if (pn) pn[1] = (unsigned short)tn;
                }
                while (top > 0) {
                    const int pe = stk[--top];
                    if (top > 20 || nfin > 20) { aborted = true; break; }          // assembler.pyx:1052-1057
                    const int endn = arena_get(pe, 0);
                    // checkPathForCycles (:999-1023): a path is only ever extended from a cycle-free path, so it
                    // suffices to compare its last node with its ancestors
                    bool cyc = false;
                    const int depth = arena_get(pe, 2);
                    if (pn && depth <= ASM_PATH) {
                        const uint4* pv = (const uint4*)pn;
                        const unsigned want = (unsigned)endn;
                        for (int j0 = 0; j0 < depth; j0 += 8) {
                            const uint4 v = pv[j0 >> 3];
                            const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int a0 = j0 + 2 * q;
                                cyc |= (a0 >= 1 && a0 < depth && (wd[q] & 0xFFFFu) == want) || (a0 + 1 < depth && (wd[q] >> 16) == want);
                            }
                        }
                    } else
                        for (int a = arena_get(pe, 1); a >= 0; a = arena_get(a, 1)) if (arena_get(a, 0) == endn) { cyc = true; break; }
                    if (cyc) continue;
                    const int col = colour_of(endn);
                    if (col == 3) { S.task_fin[t * ASM_MAX_FIN + nfin] = pe; ++nfin; }
                    else if (col == 1) continue;
                    else {
                        int end[4]; bool heavy[4];
                        const int ne = edges_of(endn, end, heavy);
                        if (pn && depth < ASM_PATH) pn[depth] = (unsigned short)endn;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {                              // assembler.pyx:1091-1107
                            if (i >= ne || overflow) continue;
                            const int c2 = colour_of(end[i]);
                            if (heavy[i] || c2 == 3 || c2 == 1) {
                                const int na = atomicAdd(&s_pool, 1);
                                if (na >= ASM_POOL) { overflow = true; continue; }
                                arena_set(na, end[i], pe, depth + 1);
                                stk[top++] = na;
                            }
                        }
                        if (overflow) break;
                    }
                }
                if (overflow) s_err = PLAT_ERR_OVERFLOW;
                S.task_nfin[t] = aborted ? 0 : nfin;
                if (fast) s_tnfin[t] = aborted ? 0 : nfin;
            }
            sync_phase(); ASM_FRESH();
            ASM_TICK(7);
            break;
        }
        sync_phase(); ASM_FRESH();

        // ---- phase G/H: variants in emission order, then the stable sort of sorted(theVars) (assembler.pyx:1476)
        int32_t* vp = var_pos + (size_t)g * P.max_vars; int32_t* vr = var_nrem + (size_t)g * P.max_vars;
        int32_t* va = var_nadd + (size_t)g * P.max_vars; int32_t* vo = var_off + (size_t)g * P.max_vars;
        uint8_t* vb = var_blob + (size_t)g * P.blob_per_region;
        // extractVarFromBubblePath (:1196-1291) for the finished path whose last element is `last`: false = no variant (:1213-1218);
        // else the path's bytes (first byte of every node) in pathb[0, plen) and the trimmed variant: position s, rl reference bytes at
        // r, al path bytes from pathb[ao]  (suffix trim first, :1253, then prefix trim advancing the position, :1262-1270)
        auto extract = [&](int t, int last, uint8_t* pathb, int& s, int& rl, int& al, int& ao, const uint8_t*& r) -> bool {
            const int plen = arena_get(last, 2);
            const int startn = S.task_node[t], endn = arena_get(last, 0);
            const unsigned fs = first_of(startn), fe = first_of(endn);
            s = refStart + (int)(fs >> 1) + (int)(fs & 1u);
            const int te = refStart + (int)(fe >> 1) + (int)(fe & 1u);
            if (te < s) return false;
            rl = te - s + 1; al = plen;
            r = ref + (s - refStart);
            int e = last;
            for (int d = plen - 1; d >= 0; --d) { pathb[d] = asm_ptr(ref, rseq, S.rep[arena_get(e, 0)])[0]; e = arena_get(e, 1); }
            while (al > 0 && rl > 0 && r[rl - 1] == pathb[al - 1]) { --rl; --al; }
            ao = 0;
            while (al > 0 && rl > 0 && r[0] == pathb[ao]) { ++r; ++ao; --rl; --al; ++s; }
            return true;
        };
        bool serialG = true;
        if (fast) {
            // one thread per finished path: the paths in emission order (tasks in order, each task's paths in the order they finished),
            // their bytes side by side in the slice (the DFS stack of the cycle check is free), output slots from scans over the paths
            const int err0 = s_err;
            const int nTasks = err0 == 0 ? s_ntasks : 0;
            const int mynf = tid < nTasks ? s_tnfin[tid] : 0;                 // (ASM_MAX_TASKS <= threads)
            int NP;
            const int pbase = asm_block_exscan(mynf, s_wsum, NP);
            int* plast = S.stack; int* ptask = S.stack + NP;
            for (int f = 0; f < mynf; ++f) { plast[pbase + f] = S.task_fin[tid * ASM_MAX_FIN + f]; ptask[pbase + f] = tid; }
            asm_sync_wg(); ASM_FRESH();
            uint8_t* pbytes = (uint8_t*)(S.stack + 2 * NP);
            const long long pcap = (long long)asm_stack_ints(P.max_pos) * 4 - 8ll * NP;
            int nvBase = 0, blobBase = 0, err = err0;
            bool fallback = false;
            // one chunk of paths: `scan(value, total)` = exclusive prefix over the chunk's threads.  A region has a dozen finished paths: up to 64
            // are the first wave's alone, with shuffles -- no workgroup-wide scans, no barriers.
            auto chunk = [&](int c0, auto&& scan) -> bool {                      // false: stop (fallback or overflow)
                const int p = c0 + tid;
                int t = 0, last = 0, plen = 0;
                if (p < NP) { last = plast[p]; t = ptask[p]; plen = arena_get(last, 2); }
                int tot;
                const int off = scan(plen, tot);
                if ((long long)tot > pcap || (P.debug & 8)) { fallback = true; return false; }     // (never seen: a path is tens of nodes; debug 8: tests)
                int s = 0, rl = 0, al = 0, ao = 0;
                const uint8_t* r = ref;
                const bool valid = p < NP && extract(t, last, pbytes + off, s, rl, al, ao, r);
                int totv, totb;
                const int vi = scan(valid ? 1 : 0, totv);
                const int bo = scan(valid ? rl + al : 0, totb);
                if (nvBase + totv > P.max_vars || blobBase + totb > P.blob_per_region) { err = PLAT_ERR_OVERFLOW; return false; }
                if (valid) {
                    const int nv = nvBase + vi, blob = blobBase + bo;
                    vp[nv] = s > 0 ? s : 0;                                        // variant.pyx:121
                    vr[nv] = rl; va[nv] = al; vo[nv] = blob;
                    for (int i = 0; i < rl; ++i) vb[blob + i] = r[i];
                    for (int i = 0; i < al; ++i) vb[blob + rl + i] = pbytes[off + ao + i];
                }
                nvBase += totv; blobBase += totb;
                return true;
            };
            if (NP <= 64) {
                if (tid < 64)
                    chunk(0, [&](int v, int& tot) -> int {
                        int x = v;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if ((tid & 63) >= d) x += y; }
                        tot = __shfl(x, 63);
                        return x - v;
                    });
            } else
                for (int c0 = 0; c0 < NP; c0 += nthr) {
                    if (!chunk(c0, [&](int v, int& tot) -> int { return asm_block_exscan(v, s_wsum, tot); })) break;
                    __syncthreads();                                              // (the bytes of this chunk are done with)
                }
            asm_sync_wg(); ASM_FRESH();
            if (!fallback) {
                serialG = false;
                if (tid == 0) { s_nv = err ? 0 : nvBase; s_err = err; }
            }
        }
        if (serialG && tid == 0) {
            int nv = 0, blob = 0, err = s_err;
            const int nTasks = (err == 0 && !(P.no_cycles && s_cycle)) ? s_ntasks : 0;
            for (int t = 0; t < nTasks && err == 0; ++t) {
                for (int f = 0; f < S.task_nfin[t] && err == 0; ++f) {
                    int s, rl, al, ao;
                    const uint8_t* r;
                    uint8_t* pathb = (uint8_t*)S.stack;
                    if (!extract(t, S.task_fin[t * ASM_MAX_FIN + f], pathb, s, rl, al, ao, r)) continue;
                    if (nv >= P.max_vars || blob + rl + al > P.blob_per_region) { err = PLAT_ERR_OVERFLOW; break; }
                    vp[nv] = s > 0 ? s : 0;                                        // variant.pyx:121
                    vr[nv] = rl; va[nv] = al; vo[nv] = blob;
                    for (int i = 0; i < rl; ++i) vb[blob + i] = r[i];
                    for (int i = 0; i < al; ++i) vb[blob + rl + i] = pathb[ao + i];
                    blob += rl + al; ++nv;
                }
            }
            s_nv = err ? 0 : nv; s_err = err;
        }
        sync_phase(); ASM_FRESH();
        if (tid == 0) {
            const int nv = s_nv, err = s_err;
            // stable insertion sort by (pos, varType, nRemoved)  (Variant.__richcmp__ '<', variant.pyx:282-363)
            auto vtype = [&](int i) { const int a = va[i], rr = vr[i]; return rr == a ? (a == 1 ? 0 : 1) : (rr == 0 ? 2 : (a == 0 ? 3 : 4)); };
            for (int i = 1; i < nv; ++i) {
                const int p0 = vp[i], r0 = vr[i], a0 = va[i], o0 = vo[i], t0 = vtype(i);
                int j = i - 1;
                while (j >= 0 && (vp[j] > p0 || (vp[j] == p0 && (vtype(j) > t0 || (vtype(j) == t0 && vr[j] > r0))))) {
                    vp[j + 1] = vp[j]; vr[j + 1] = vr[j]; va[j + 1] = va[j]; vo[j + 1] = vo[j]; --j;
                }
                vp[j + 1] = p0; vr[j + 1] = r0; va[j + 1] = a0; vo[j + 1] = o0;
            }
            var_count[g] = err ? 0 : nv;
            status[g] = err;
            if (err) s_dirty = 1;                            // (a region that gave up did not tidy up behind itself)
        }
        if (s_lds == 0)                                   // a region done on the global path has written successor bytes of low node ids
            for (int i = tid; i < ASM_LDS_NODES * ASM_MAX_SUCC; i += nthr) S.succ_c[i] = 0;
        sync_phase(); ASM_FRESH();
        ASM_TICK(8);
        if (work) {
            if (tid == 0) s_next_g = (int)gridDim.x + (int)atomicAdd((unsigned long long*)work, 1ull);
            __syncthreads();
            g = s_next_g;
        } else g += gridDim.x;
    }
    if (wg_sig && tid == 0 && s_dirty == 0) wg_sig[blockIdx.x] = sig;      // (behind the last region's barrier: every store of this workgroup has been issued)
}

__global__ void k_asm_sizes(plat_assembly_batch b, long long* out /* [0]=max ref, [1]=max reads, [2]=max positions, [3]=err */)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= b.n_regions) return;
    const long long rl = b.ref_off[g + 1] - b.ref_off[g];
    const int r0 = b.reg_read_begin[g], r1 = b.reg_read_begin[g + 1];
    if (rl < 0 || rl >= 0x40000000ll || r1 < r0) { out[3] = PLAT_ERR_BAD_INPUT; return; }
    long long pos = rl + 2;
    const long long bytes = r1 > r0 ? b.read_off[r1] - b.read_off[r0] : 0;
    if (bytes < 0 || bytes >= 0x3FFFFFFFll) { out[3] = PLAT_ERR_BAD_INPUT; return; }
    pos += bytes + 2 * (long long)(r1 - r0);
    atomicMax((unsigned long long*)&out[0], (unsigned long long)rl);
    atomicMax((unsigned long long*)&out[1], (unsigned long long)(r1 - r0));
    atomicMax((unsigned long long*)&out[2], (unsigned long long)pos);
}

// the caller's sizes against the batch (plat_assemble_batch_async): out[3] = 0 or the error every tile is refused with
__global__ void k_asm_check(plat_assembly_batch b, long long max_ref, long long max_reads, long long max_pos, long long* out)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= b.n_regions) return;
    const long long rl = b.ref_off[g + 1] - b.ref_off[g];
    const int r0 = b.reg_read_begin[g], r1 = b.reg_read_begin[g + 1];
    if (rl < 0 || rl >= 0x40000000ll || r1 < r0) { out[3] = PLAT_ERR_BAD_INPUT; return; }
    const long long bytes = r1 > r0 ? b.read_off[r1] - b.read_off[r0] : 0;
    if (bytes < 0 || bytes >= 0x3FFFFFFFll) { out[3] = PLAT_ERR_BAD_INPUT; return; }
    if (rl > max_ref || r1 - r0 > max_reads || rl + 2 + bytes + 2 * (long long)(r1 - r0) > max_pos) out[3] = PLAT_ERR_BAD_HINTS;
}

}  // namespace plat

using namespace plat;

static int asm_launch(plat_ctx* ctx, const plat_assembly_batch& b, int kmer_size, int min_qual, int min_weight, int no_cycles,
                      int max_vars_per_region, int blob_per_region, int max_ref, int max_reads, long long max_pos_raw,
                      int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd, int32_t* var_off, uint8_t* var_blob,
                      int32_t* status, const long long* verdict, long long* work, hipStream_t st);

PLAT_EXPORT int plat_assemble_batch(plat_ctx* ctx, const plat_assembly_batch* batch, int kmer_size, int min_qual,
                                    int min_weight, int no_cycles, int max_vars_per_region, int blob_per_region,
                                    int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd,
                                    int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream)
{
    if (!ctx || !batch) return PLAT_ERR_INVALID;
    const plat_assembly_batch b = *batch;
    if (b.n_regions < 0 || b.n_reads < 0 || kmer_size < 5 || kmer_size > 200 || max_vars_per_region <= 0 ||
        blob_per_region <= 0)
        return PLAT_ERR_INVALID;
    if (b.n_regions == 0) return PLAT_OK;
    if (!b.ref_seq || !b.ref_off || !b.ref_start || !b.assem_start || !b.assem_end || !b.reg_read_begin ||
        !b.read_off || !var_count || !var_pos || !var_nrem || !var_nadd || !var_off || !var_blob || !status)
        return PLAT_ERR_INVALID;
    if (b.n_reads > 0 && (!b.read_seq || !b.read_qual)) return PLAT_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = plat_reserve(ctx, ctx->counters, 64 * sizeof(long long));
    if (rc) return rc;
    long long* d_sz = (long long*)ctx->counters.ptr;
    PLAT_HIP(ctx, hipMemsetAsync(d_sz, 0, 8 * sizeof(long long), st));
    hipLaunchKernelGGL(k_asm_sizes, dim3((b.n_regions + 255) / 256), dim3(256), 0, st, b, d_sz);
    int64_t* hb = ctx->h_readback;
    PLAT_HIP(ctx, hipMemcpyAsync(hb, d_sz, 4 * sizeof(long long), hipMemcpyDeviceToHost, st));
    PLAT_HIP(ctx, hipStreamSynchronize(st));
    if (hb[3] != 0) return (int)hb[3];
    return asm_launch(ctx, b, kmer_size, min_qual, min_weight, no_cycles, max_vars_per_region, blob_per_region, (int)hb[0], (int)hb[1], hb[2],
                      var_count, var_pos, var_nrem, var_nadd, var_off, var_blob, status, nullptr, d_sz + 4, st);
}

PLAT_EXPORT int plat_assemble_batch_async(plat_ctx* ctx, const plat_assembly_batch* batch, const plat_assembly_hints* hints, int kmer_size, int min_qual,
                                          int min_weight, int no_cycles, int max_vars_per_region, int blob_per_region,
                                          int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd,
                                          int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream)
{
    if (!ctx || !batch || !hints) return PLAT_ERR_INVALID;
    const plat_assembly_batch b = *batch;
    if (b.n_regions < 0 || b.n_reads < 0 || kmer_size < 5 || kmer_size > 200 || max_vars_per_region <= 0 || blob_per_region <= 0 ||
        hints->max_ref_len < 0 || hints->max_reads_per_region < 0 || hints->max_positions < 0 || hints->max_positions >= 0x3FFFFFF0ll)
        return PLAT_ERR_INVALID;
    if (b.n_regions == 0) return PLAT_OK;
    if (!b.ref_seq || !b.ref_off || !b.ref_start || !b.assem_start || !b.assem_end || !b.reg_read_begin ||
        !b.read_off || !var_count || !var_pos || !var_nrem || !var_nadd || !var_off || !var_blob || !status)
        return PLAT_ERR_INVALID;
    if (b.n_reads > 0 && (!b.read_seq || !b.read_qual)) return PLAT_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = plat_reserve(ctx, ctx->counters, 64 * sizeof(long long));
    if (rc) return rc;
    long long* d_sz = (long long*)ctx->counters.ptr;
    PLAT_HIP(ctx, hipMemsetAsync(d_sz, 0, 8 * sizeof(long long), st));
    hipLaunchKernelGGL(k_asm_check, dim3((b.n_regions + 255) / 256), dim3(256), 0, st, b, (long long)hints->max_ref_len, (long long)hints->max_reads_per_region,
                       (long long)hints->max_positions, d_sz);
    return asm_launch(ctx, b, kmer_size, min_qual, min_weight, no_cycles, max_vars_per_region, blob_per_region, hints->max_ref_len, hints->max_reads_per_region,
                      hints->max_positions, var_count, var_pos, var_nrem, var_nadd, var_off, var_blob, status, d_sz, d_sz + 4, st);
}

static int asm_launch(plat_ctx* ctx, const plat_assembly_batch& b, int kmer_size, int min_qual, int min_weight, int no_cycles,
                      int max_vars_per_region, int blob_per_region, int max_ref, int max_reads, long long max_pos_raw,
                      int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd, int32_t* var_off, uint8_t* var_blob,
                      int32_t* status, const long long* verdict, long long* work, hipStream_t st)
{
    int rc;
    // k may grow to 55 under noCycles, which only lowers the number of edges: size for the initial k
    long long max_pos = max_pos_raw + 16;
    if (max_pos > 0x3FFFFFFFll) return PLAT_ERR_OVERFLOW;
    int cap = 1024;
    while ((long long)cap * 3 < max_pos * 4 * 2) cap <<= 1;     // load factor <= 3/8 even if every occurrence were distinct
    AsmParams P;
    P.kmer = kmer_size; P.min_qual = min_qual; P.min_weight = min_weight; P.no_cycles = no_cycles;
    P.max_vars = max_vars_per_region; P.blob_per_region = blob_per_region; P.cap = cap; P.max_pos = (int)max_pos;
    P.timing = getenv("PLAT_ASM_TIMING") != nullptr;
    { const char* ef = getenv("PLAT_ASM_FUSED"); P.fused = !(ef && ef[0] == '0'); }
    { const char* ed = getenv("PLAT_ASM_DEBUG"); P.debug = ed ? atoi(ed) : 0; }
    const size_t per_block = asm_scratch_bytes(cap, (int)max_pos, max_ref, max_reads);
    P.scratch_per_block = (long long)per_block;
    // the kernel is bound by the latency of dependent L2 accesses, not by bandwidth or issue: one region per CU at a time (its graph takes
    // most of the CU's LDS) and as the scratch memory allows (PLAT_ASM_WG_PER_CU overrides, for measurements)
    int per_cu = 1;                                  // (the graph of a region takes most of a CU's LDS)
    if (const char* e = getenv("PLAT_ASM_WG_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    int nblk = b.n_regions < per_cu * ctx->n_cu ? b.n_regions : per_cu * ctx->n_cu;
    while (nblk > 1 && per_block * (size_t)nblk > ((size_t)96 << 30)) nblk = nblk * 3 / 4;
    {
        const void* before = ctx->asm_scratch.ptr; const size_t cap0 = ctx->asm_scratch.cap;
        if ((rc = plat_reserve(ctx, ctx->asm_scratch, per_block * (size_t)nblk))) return rc;
        if (ctx->asm_scratch.ptr != before || ctx->asm_scratch.cap != cap0) ++ctx->asm_epoch;
    }
    if (!ctx->asm_sig.ptr) {
        if ((rc = plat_reserve(ctx, ctx->asm_sig, 4096 * sizeof(unsigned long long)))) return rc;
        PLAT_HIP(ctx, hipMemsetAsync(ctx->asm_sig.ptr, 0, 4096 * sizeof(unsigned long long), st));
    }
    unsigned long long* wg_sig = (nblk <= 4096 && !getenv("PLAT_ASM_NO_KEEP")) ? (unsigned long long*)ctx->asm_sig.ptr : nullptr;
    // A launch only rewrites wg_sig[] of the workgroups it runs, and a slice sits at blockIdx.x * per_block: a launch with ANOTHER layout (or one
    // that ran without the signature array) overwrites slices of workgroups whose stored signature it never touches.  Every stored signature is
    // therefore made stale -- the epoch is part of it and only grows -- whenever the layout differs from the previous launch's.
    {
        unsigned long long lay = 0xCBF29CE484222325ull;
        for (unsigned long long v : {(unsigned long long)per_block, (unsigned long long)cap, (unsigned long long)max_pos, (unsigned long long)max_ref, (unsigned long long)max_reads})
            lay = (lay ^ v) * 0x100000001B3ull + (lay >> 31);
        lay |= 1ull;
        if (lay != ctx->asm_last_layout || !ctx->asm_last_kept) ++ctx->asm_epoch;
        ctx->asm_last_layout = lay;
        ctx->asm_last_kept = wg_sig != nullptr;
    }
    // the layout of a workgroup's slice: what asm_carve is given + where the slices lie (never 0)
    unsigned long long sig = 0x9E3779B97F4A7C15ull;
    for (unsigned long long v : {(unsigned long long)(uintptr_t)ctx->asm_scratch.ptr, (unsigned long long)per_block, (unsigned long long)cap, (unsigned long long)max_pos,
                                 (unsigned long long)max_ref, (unsigned long long)max_reads, ctx->asm_epoch})
        sig = (sig ^ v) * 0x100000001B3ull + (sig >> 29);
    sig |= 1ull;
    const int lds_bytes = ASM_LDS_BYTES;
    PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    { PLAT_KT_BEGIN(ctx, PLAT_KT_ASSEMBLE, st); hipLaunchKernelGGL(k_assemble, dim3(nblk), dim3(ASM_THREADS), lds_bytes, st, b, P, (char*)ctx->asm_scratch.ptr, max_ref, max_reads,
                       var_count, var_pos, var_nrem, var_nadd, var_off, var_blob, status, verdict, getenv("PLAT_ASM_STATIC") ? nullptr : work, wg_sig, sig); PLAT_KT_END(ctx, PLAT_KT_ASSEMBLE, st); }   // (PLAT_ASM_STATIC: tile g on workgroup g % grid, for A/B runs)
    PLAT_HIP(ctx, hipGetLastError());
    if (P.timing) {
        unsigned long long t[16];
        PLAT_HIP(ctx, hipStreamSynchronize(st));
        PLAT_HIP(ctx, hipMemcpyFromSymbol(t, HIP_SYMBOL(g_asm_ticks), sizeof t));
        fprintf(stderr, "k_assemble, 10 ns ticks per phase summed over %d workgroups (ticket scan, A insert, B ids, C events, D successors, cycles, E starts, F paths, G variants):", nblk);
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", t[i]);       // (11..15, builds with -DPLAT_ASM_SECTIONS: the first wave's shader-clock cycles in the read loop -- rest, validity, gather, probes, events)
        fprintf(stderr, "\n");
        memset(t, 0, sizeof t);
        PLAT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_asm_ticks), t, sizeof t));
    }
    return PLAT_OK;
}
