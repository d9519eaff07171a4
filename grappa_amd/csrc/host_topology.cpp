// libgrappa_host.so: host-side graph preparation (include/grappa_host.h) -- tuple enumeration and ring / degree features in O(atoms).
#include "../../include/grappa_host.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

constexpr int OK = 0, ERR_ARG = -1, ERR_WORKSPACE = -3;   // include/grappa_hip.h GRAPPA_OK / GRAPPA_ERR_ARG / GRAPPA_ERR_WORKSPACE
constexpr int MAX_RING = 8;

// CSR adjacency over ids 0..n-1; every bond appears in both lists (duplicates kept, as the reference's dict of lists keeps them)
struct Adjacency {
    std::vector<int64_t> ptr;
    std::vector<int32_t> nbr;
    void build(int n, int n_bonds, const int32_t* bonds, bool sort_lists) {
        ptr.assign((size_t)n + 1, 0);
        for (int b = 0; b < n_bonds; ++b) {
            ++ptr[bonds[2 * b] + 1];
            ++ptr[bonds[2 * b + 1] + 1];
        }
        for (int i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
        nbr.resize((size_t)ptr[n]);
        std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
        for (int b = 0; b < n_bonds; ++b) {
            const int32_t u = bonds[2 * b], v = bonds[2 * b + 1];
            nbr[fill[u]++] = v;
            nbr[fill[v]++] = u;
        }
        if (sort_lists)
            for (int i = 0; i < n; ++i) std::sort(nbr.begin() + ptr[i], nbr.begin() + ptr[i + 1]);
    }
    const int32_t* begin(int v) const { return nbr.data() + ptr[v]; }
    const int32_t* end(int v) const { return nbr.data() + ptr[v + 1]; }
};

// dynamic bitset over the edges of one ring system
struct Bits {
    std::vector<uint64_t> w;
    explicit Bits(int nbits = 0) : w((size_t)(nbits + 63) / 64, 0) {}
    void flip(int i) { w[i >> 6] ^= 1ull << (i & 63); }
    int top() const {
        for (int k = (int)w.size() - 1; k >= 0; --k)
            if (w[k]) return k * 64 + 63 - __builtin_clzll(w[k]);
        return -1;
    }
    void operator^=(const Bits& o) {
        for (size_t k = 0; k < w.size(); ++k) w[k] ^= o.w[k];
    }
    bool operator<(const Bits& o) const { return w < o.w; }
    bool operator==(const Bits& o) const { return w == o.w; }
};

struct Cycle {
    int len;
    Bits bits;
    int atoms[MAX_RING];
};

}  // namespace

extern "C" int grappa_host_abi_version(void) { return 3; }

extern "C" int grappa_topo_enumerate(int n_bonds, const int32_t* bonds, int32_t* angles, int64_t cap_angles, int32_t* propers,
                                     int64_t cap_propers, int64_t* n_angles, int64_t* n_propers) {
    if (n_bonds < 0 || (n_bonds > 0 && !bonds) || !n_angles || !n_propers) return ERR_ARG;
    int32_t max_id = -1;
    for (int b = 0; b < n_bonds; ++b) {
        const int32_t u = bonds[2 * b], v = bonds[2 * b + 1];
        if (u < 0 || v < 0 || u == v) return ERR_ARG;                    // the reference asserts on self-bonds
        max_id = std::max(max_id, std::max(u, v));
    }
    const int n = max_id + 1;
    // atoms in order of first appearance in the bond list (the iteration order of the reference's neighbour dict)
    std::vector<int32_t> order;
    order.reserve((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    for (int b = 0; b < n_bonds; ++b)
        for (int e = 0; e < 2; ++e) {
            const int32_t a = bonds[2 * b + e];
            if (!seen[a]) {
                seen[a] = 1;
                order.push_back(a);
            }
        }
    Adjacency adj;
    adj.build(n, n_bonds, bonds, true);
    int64_t na = 0, np = 0;
    bool overflow = false;
    for (const int32_t a1 : order)
        for (const int32_t* p2 = adj.begin(a1); p2 != adj.end(a1); ++p2) {
            const int32_t a2 = *p2;
            for (const int32_t* p3 = adj.begin(a2); p3 != adj.end(a2); ++p3) {
                const int32_t a3 = *p3;
                if (a3 == a1) continue;
                if (a1 < a3) {
                    if (angles) {
                        if (na < cap_angles) {
                            int32_t* r = angles + 3 * na;
                            r[0] = a1; r[1] = a2; r[2] = a3;
                        } else {
                            overflow = true;
                        }
                    }
                    ++na;
                }
                for (const int32_t* p4 = adj.begin(a3); p4 != adj.end(a3); ++p4) {
                    const int32_t a4 = *p4;
                    if (a4 >= a1) break;                                    // neighbour lists ascend: nothing smaller follows
                    if (a4 == a2) continue;
                    if (propers) {
                        if (np < cap_propers) {
                            int32_t* r = propers + 4 * np;
                            r[0] = a4; r[1] = a3; r[2] = a2; r[3] = a1;
                        } else {
                            overflow = true;
                        }
                    }
                    ++np;
                }
            }
        }
    *n_angles = na;
    *n_propers = np;
    return overflow ? ERR_WORKSPACE : OK;
}

extern "C" int grappa_degree_encoding(int n_atoms, int n_bonds, const int32_t* bonds, float* enc) {
    if (n_atoms < 0 || n_bonds < 0 || (n_bonds > 0 && !bonds) || (n_atoms > 0 && !enc)) return ERR_ARG;
    std::vector<int32_t> deg((size_t)n_atoms, 0);
    for (int b = 0; b < 2 * n_bonds; ++b) {
        if (bonds[b] < 0 || bonds[b] >= n_atoms) return ERR_ARG;
        ++deg[bonds[b]];
    }
    std::memset(enc, 0, sizeof(float) * 6 * (size_t)n_atoms);
    for (int i = 0; i < n_atoms; ++i)
        if (deg[i] >= 1 && deg[i] <= 6) enc[6 * (size_t)i + deg[i] - 1] = 1.0f;
    return OK;
}

extern "C" int grappa_ring_encoding(int n_atoms, int n_bonds, const int32_t* bonds, float* enc) {
    if (n_atoms < 0 || n_bonds < 0 || (n_bonds > 0 && !bonds) || (n_atoms > 0 && !enc)) return ERR_ARG;
    for (int b = 0; b < 2 * n_bonds; ++b)
        if (bonds[b] < 0 || bonds[b] >= n_atoms) return ERR_ARG;
    std::memset(enc, 0, sizeof(float) * 7 * (size_t)n_atoms);
    Adjacency adj;
    adj.build(n_atoms, n_bonds, bonds, true);
    // ---- bridges (iterative lowlink DFS; parallel edges to the DFS parent are ignored, as one bond) -> atoms on a cycle
    std::vector<int32_t> disc((size_t)n_atoms, -1), low((size_t)n_atoms, 0), parent((size_t)n_atoms, -1);
    std::vector<int64_t> it((size_t)n_atoms, 0);
    std::vector<char> in_ring((size_t)n_atoms, 0), tree_bridge((size_t)n_atoms, 1);   // tree_bridge[v]: edge (parent[v], v) is a bridge
    std::vector<int32_t> stack;
    int32_t timer = 0;
    for (int root = 0; root < n_atoms; ++root) {
        if (disc[root] != -1) continue;
        disc[root] = low[root] = timer++;
        it[root] = adj.ptr[root];
        stack.push_back(root);
        while (!stack.empty()) {
            const int32_t v = stack.back();
            if (it[v] < adj.ptr[v + 1]) {
                const int32_t w = adj.nbr[it[v]++];
                if (w == parent[v]) continue;
                if (disc[w] == -1) {
                    disc[w] = low[w] = timer++;
                    parent[w] = v;
                    it[w] = adj.ptr[w];
                    stack.push_back(w);
                } else {
                    low[v] = std::min(low[v], disc[w]);
                }
            } else {
                stack.pop_back();
                const int32_t p = parent[v];
                if (p != -1) {
                    low[p] = std::min(low[p], low[v]);
                    if (low[v] <= disc[p]) {                                  // (p, v) lies on a cycle
                        tree_bridge[v] = 0;
                        in_ring[v] = in_ring[p] = 1;
                    }
                }
            }
        }
    }
    auto is_ring_edge = [&](int32_t a, int32_t b) {
        if (!in_ring[a] || !in_ring[b]) return false;
        if (parent[b] == a) return !tree_bridge[b];
        if (parent[a] == b) return !tree_bridge[a];
        return true;                                                           // a non-tree edge always closes a cycle
    };
    // ---- ring systems = connected components over ring edges; local edge numbering per system
    std::vector<int32_t> comp((size_t)n_atoms, -1);
    std::vector<int32_t> members, queue;
    std::vector<char> on_path((size_t)n_atoms, 0);     // every mark is cleared when its frame pops
    for (int s = 0; s < n_atoms; ++s) {
        enc[7 * (size_t)s] = in_ring[s] ? 1.0f : 0.0f;
        if (!in_ring[s] || comp[s] != -1) continue;
        members.clear();
        queue.assign(1, s);
        comp[s] = s;
        while (!queue.empty()) {
            const int32_t v = queue.back();
            queue.pop_back();
            members.push_back(v);
            for (const int32_t* p = adj.begin(v); p != adj.end(v); ++p)
                if (comp[*p] == -1 && is_ring_edge(v, *p)) {
                    comp[*p] = s;
                    queue.push_back(*p);
                }
        }
        std::sort(members.begin(), members.end());
        std::unordered_map<uint64_t, int> edge_id;
        for (const int32_t a : members)
            for (const int32_t* p = adj.begin(a); p != adj.end(a); ++p)
                if (a < *p && is_ring_edge(a, *p)) edge_id.emplace(((uint64_t)(uint32_t)a << 32) | (uint32_t)*p, (int)edge_id.size());
        const int ne = (int)edge_id.size();
        auto eid = [&](int32_t a, int32_t b) {
            if (a > b) std::swap(a, b);
            return edge_id.at(((uint64_t)(uint32_t)a << 32) | (uint32_t)b);
        };
        // ---- every simple cycle of length 3..8 once: found from its smallest atom, orientation fixed by path[1] < path[last]
        std::vector<Cycle> cycles;
        int path[MAX_RING];
        struct Frame { int32_t v; int64_t next; };
        std::vector<Frame> fr;
        for (const int32_t start : members) {
            int depth = 0;
            path[depth++] = start;
            on_path[start] = 1;
            fr.assign(1, Frame{start, adj.ptr[start]});
            while (!fr.empty()) {
                Frame& f = fr.back();
                if (f.next == adj.ptr[f.v + 1]) {
                    on_path[f.v] = 0;
                    --depth;
                    fr.pop_back();
                    continue;
                }
                const int32_t w = adj.nbr[f.next++];
                if (w == start && depth >= 3) {
                    if (path[1] < path[depth - 1] && is_ring_edge(f.v, w)) {
                        Cycle c;
                        c.len = depth;
                        c.bits = Bits(ne);
                        bool ok = true;
                        for (int i = 0; i < depth && ok; ++i) {
                            const int32_t a = path[i], b = path[(i + 1) % depth];
                            if (!is_ring_edge(a, b)) ok = false;
                            else c.bits.flip(eid(a, b));
                            c.atoms[i] = a;
                        }
                        if (ok) cycles.push_back(std::move(c));
                    }
                } else if (w > start && comp[w] == s && !on_path[w] && depth < MAX_RING && is_ring_edge(f.v, w)) {
                    path[depth++] = w;
                    on_path[w] = 1;
                    fr.push_back(Frame{w, adj.ptr[w]});
                }
            }
        }
        // parallel bonds can produce the same edge set twice: keep one per edge set (the reference graph has no parallel bonds)
        std::sort(cycles.begin(), cycles.end(), [](const Cycle& x, const Cycle& y) { return x.len != y.len ? x.len < y.len : x.bits < y.bits; });
        cycles.erase(std::unique(cycles.begin(), cycles.end(), [](const Cycle& x, const Cycle& y) { return x.len == y.len && x.bits == y.bits; }),
                     cycles.end());
        // ---- relevant cycles: not a GF(2) sum of strictly shorter cycles (basis grows only after a whole length class is tested)
        std::unordered_map<int, Bits> basis;
        auto reduce = [&](Bits v) {
            for (int t = v.top(); t >= 0; t = v.top()) {
                auto f = basis.find(t);
                if (f == basis.end()) break;
                v ^= f->second;
            }
            return v;
        };
        for (size_t i = 0; i < cycles.size();) {
            size_t j = i;
            while (j < cycles.size() && cycles[j].len == cycles[i].len) ++j;
            std::vector<size_t> keep;
            for (size_t c = i; c < j; ++c)
                if (reduce(cycles[c].bits).top() >= 0) keep.push_back(c);
            for (const size_t c : keep) {
                for (int a = 0; a < cycles[c].len; ++a) enc[7 * (size_t)cycles[c].atoms[a] + cycles[c].len - 2] = 1.0f;
                Bits r = reduce(cycles[c].bits);
                const int t = r.top();
                if (t >= 0) basis.emplace(t, std::move(r));
            }
            i = j;
        }
    }
    return OK;
}


// ------------------------------------------------------------------------------------------------ index plan of a batched graph
// grappa_amd/batch.py BatchPlan in O(E + sum s T) counting sorts instead of ~60 numpy calls (0.4 ms for one 40-atom molecule: a seventh of
// a recorded Grappa.predict call).  Same arrays, element for element (tests/test_host_plan.py).
extern "C" int grappa_plan_build(int N, int64_t E, const int64_t* src, const int64_t* dst, const int32_t* T, const int32_t* const* idx,
                                 int32_t* indptr, int32_t* indices, int32_t* rev, int32_t* const* inv_ptr, int32_t* const* inv_rows,
                                 int32_t* inc_ptr, int32_t* inc_code, int32_t* max_degree, int32_t* status_detail) {
    static const int ARITY[4] = {2, 3, 4, 4};
    if (N < 0 || E < 0 || (E > 0 && (!src || !dst)) || !T || !idx || !indptr || !inv_ptr || !inv_rows || !inc_ptr || !max_degree) return ERR_ARG;
    if (status_detail) *status_detail = 0;
    // ---- CSR by destination, neighbours ascending by source id, duplicates in input order (= numpy's lexsort((src, dst)))
    for (int i = 0; i <= N; ++i) indptr[i] = 0;
    for (int64_t e = 0; e < E; ++e) {
        if (src[e] < 0 || src[e] >= N || dst[e] < 0 || dst[e] >= N) return ERR_ARG;
        ++indptr[dst[e] + 1];
    }
    for (int i = 0; i < N; ++i) indptr[i + 1] += indptr[i];
    std::vector<int64_t> order((size_t)E);
    {
        // stable counting sort by src, then by dst: the result is ordered by (dst, src) with ties in input order
        std::vector<int64_t> cnt((size_t)N + 1, 0), tmp((size_t)E);
        for (int64_t e = 0; e < E; ++e) ++cnt[src[e] + 1];
        for (int i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
        for (int64_t e = 0; e < E; ++e) tmp[cnt[src[e]]++] = e;
        std::vector<int64_t> pos(indptr, indptr + N);
        for (int64_t k = 0; k < E; ++k) {
            const int64_t e = tmp[k];
            order[pos[dst[e]]++] = e;
        }
    }
    for (int64_t k = 0; k < E; ++k) indices[k] = (int32_t)src[order[k]];
    // reverse edge slot: edge k = (u -> v) lies in v's list; its reverse (v -> u) is the FIRST entry with source v in u's list
    for (int64_t k = 0; k < E; ++k) {
        const int64_t u = src[order[k]], v = dst[order[k]];
        const int32_t* lo = indices + indptr[u];
        const int32_t* hi = indices + indptr[u + 1];
        const int32_t* it = std::lower_bound(lo, hi, (int32_t)v);
        if (it == hi || *it != (int32_t)v) {
            if (status_detail) *status_detail = 1;           // the graph does not hold both directions of a bond
            return ERR_ARG;
        }
        rev[k] = (int32_t)(it - indices);
    }
    int maxdeg = 0;
    for (int i = 0; i < N; ++i) {
        const int d = indptr[i + 1] - indptr[i];
        if (d == 0) {
            if (status_detail) *status_detail = 2;           // a 0-in-degree node
            return ERR_ARG;
        }
        maxdeg = d > maxdeg ? d : maxdeg;
    }
    *max_degree = maxdeg;
    // ---- per level: inverse incidence atom -> token rows (pos * T + t), rows ascending (a stable sort of the rows by atom)
    int64_t total = 0;
    for (int l = 0; l < 4; ++l) total += (int64_t)ARITY[l] * T[l];
    for (int i = 0; i <= N; ++i) inc_ptr[i] = 0;
    for (int l = 0; l < 4; ++l) {
        const int s = ARITY[l], Tl = T[l];
        int32_t* ptr = inv_ptr[l];
        for (int i = 0; i <= N; ++i) ptr[i] = 0;
        for (int t = 0; t < Tl; ++t)
            for (int p = 0; p < s; ++p) {
                const int a = idx[l][(size_t)t * s + p];
                if (a < 0 || a >= N) return ERR_ARG;
                ++ptr[a + 1];
                ++inc_ptr[a + 1];
            }
        for (int i = 0; i < N; ++i) ptr[i + 1] += ptr[i];
        std::vector<int32_t> pos(ptr, ptr + N);
        for (int p = 0; p < s; ++p)
            for (int t = 0; t < Tl; ++t) inv_rows[l][pos[idx[l][(size_t)t * s + p]]++] = p * Tl + t;
    }
    // ---- packed incidence of the force kernel: code = (tuple << 4) | (level << 2) | pos, grouped by atom, inside an atom in the order
    // level, position, tuple (the stable sort of the concatenated lists)
    for (int i = 0; i < N; ++i) inc_ptr[i + 1] += inc_ptr[i];
    if (total > 0) {
        if (!inc_code) return ERR_ARG;
        std::vector<int32_t> pos(inc_ptr, inc_ptr + N);
        for (int l = 0; l < 4; ++l) {
            const int s = ARITY[l], Tl = T[l];
            if (Tl > 0 && (((int64_t)(Tl - 1) << 4) | 15) >= (int64_t(1) << 31)) return ERR_ARG;
            for (int p = 0; p < s; ++p)
                for (int t = 0; t < Tl; ++t) inc_code[pos[idx[l][(size_t)t * s + p]]++] = (int32_t)(((int64_t)t << 4) | (l << 2) | p);
        }
    }
    return OK;
}


// connected components of an undirected graph given as directed edges: label[a] = smallest atom index of a's component
extern "C" int grappa_components(int N, int64_t E, const int64_t* src, const int64_t* dst, int32_t* label) {
    if (N < 0 || E < 0 || (E > 0 && (!src || !dst)) || (N > 0 && !label)) return ERR_ARG;
    std::vector<int32_t> parent((size_t)N);
    for (int i = 0; i < N; ++i) parent[i] = i;
    auto find = [&](int a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    for (int64_t e = 0; e < E; ++e) {
        if (src[e] < 0 || src[e] >= N || dst[e] < 0 || dst[e] >= N) return ERR_ARG;
        const int ra = find((int)src[e]), rb = find((int)dst[e]);
        if (ra != rb) parent[ra > rb ? ra : rb] = ra > rb ? rb : ra;          // the smaller index stays the root
    }
    for (int i = 0; i < N; ++i) label[i] = find(i);
    return OK;
}


// index tables of the (atom, position) formulation of a writer's first layer (grappa_amd/batch.py _position_tables), one flat int32 array:
//   idx_id (N x s) | invid_ptr (N + 1) | invid_rows (N s) | idx_tab (T x s) | invtab_ptr (s N + 1) | invtab_rows (s T)
// every part starting at the next multiple of 4 elements; returns the number of elements written (or needed, if out == NULL), < 0 on error
extern "C" long long grappa_position_tables(int N, int T, int s, const int32_t* idx, int32_t* out, long long out_len) {
    if (N < 0 || T < 0 || s < 1 || s > 4 || (T > 0 && !idx)) return ERR_ARG;
    const long long sizes[6] = {(long long)N * s, (long long)N + 1, (long long)N * s, (long long)T * s, (long long)s * N + 1, (long long)s * T};
    long long offs[7];
    offs[0] = 0;
    for (int i = 0; i < 6; ++i) offs[i + 1] = offs[i] + (sizes[i] + 3) / 4 * 4;
    const long long need = offs[6] > 4 ? offs[6] : 4;
    if (!out) return need;
    if (out_len < need) return ERR_ARG;
    for (long long i = 0; i < need; ++i) out[i] = 0;
    int32_t* idx_id = out + offs[0];
    int32_t* invid_ptr = out + offs[1];
    int32_t* invid_rows = out + offs[2];
    int32_t* idx_tab = out + offs[3];
    int32_t* invtab_ptr = out + offs[4];
    int32_t* invtab_rows = out + offs[5];
    for (int n = 0; n < N; ++n)
        for (int p = 0; p < s; ++p) {
            idx_id[(long long)n * s + p] = n;
            invid_rows[(long long)n * s + p] = p * N + n;
        }
    for (int n = 0; n <= N; ++n) invid_ptr[n] = n * s;
    for (long long t = 0; t < T; ++t)
        for (int p = 0; p < s; ++p) {
            const int32_t a = idx[t * s + p];
            if (a < 0 || a >= N) return ERR_ARG;
            const int32_t row = p * N + a;
            idx_tab[t * s + p] = row;
            ++invtab_ptr[row + 1];
        }
    for (long long r = 0; r < (long long)s * N; ++r) invtab_ptr[r + 1] += invtab_ptr[r];
    std::vector<int32_t> fill(invtab_ptr, invtab_ptr + (long long)s * N);
    for (int p = 0; p < s; ++p)                                   // token rows pos * T + t in ascending order: a stable sort by table row
        for (long long t = 0; t < T; ++t) invtab_rows[fill[idx_tab[t * s + p]]++] = (int32_t)(p * (long long)T + t);
    return need;
}
