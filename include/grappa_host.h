/* libgrappa_host.so -- native HOST-side graph preparation for the Grappa hot path.
 *
 * The reference prepares a molecule's interaction tuples and graph features in pure Python / RDKit on the CPU
 * (utils/tuple_indices.py, utils/rdkit_utils.py); at 50 k atoms that takes seconds per molecule.  These entry points do the same
 * work in O(atoms) C++ on the host, before the graph is copied to HBM.  Plain C ABI: int status (0 = ok, negative = error,
 * same codes as include/grappa_hip.h), caller-owned buffers, no global state, re-entrant.
 *
 * Reference-side binding a maintainer would add: INTEGRATION.md ("ctypes stub for libgrappa_host.so").
 */
#ifndef GRAPPA_HOST_H
#define GRAPPA_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int grappa_host_abi_version(void);

/* Angles and proper torsions of a bond graph: replaces get_idx_tuples + get_neighbor_dict (utils/tuple_indices.py:7-63, :66-83).
 * Same SETS and the same ROW ORDER as the reference (the row order of the tuple tables is the row order of the parameters
 * Grappa.predict returns): atoms are visited in order of first appearance in `bonds`, neighbour lists ascend;
 * angle (a, b, c) is emitted with a < c, proper (a, b, c, d) with a < d.
 *   bonds     [n_bonds][2] atom ids (any non-negative int32; a self-bond is an error, as in the reference)
 *   angles    [cap_angles][3]  or NULL (count only);  propers [cap_propers][4] or NULL
 *   n_angles / n_propers: rows the graph has (always written).  Returns GRAPPA_ERR_WORKSPACE if a capacity is too small. */
int grappa_topo_enumerate(int n_bonds, const int32_t* bonds, int32_t* angles, int64_t cap_angles, int32_t* propers,
                          int64_t cap_propers, int64_t* n_angles, int64_t* n_propers);

/* One-hot of the number of bonded neighbours 1..6: enc [n_atoms][6] (utils/rdkit_utils.py:55-67 get_degree on a graph built
 * from bonds only, :27-52).  bonds hold atom INDICES 0..n_atoms-1. */
int grappa_degree_encoding(int n_atoms, int n_bonds, const int32_t* bonds, float* enc);

/* Ring membership: enc [n_atoms][7] = [in any ring, in a ring of size 3, 4, 5, 6, 7, 8] (utils/rdkit_utils.py:7-24
 * get_ring_encoding: atom.IsInRing(), atom.IsInRingSize(3..8)).  RDKit is not available offline; the ring set used here is the
 * set of relevant cycles of length <= 8 (cycles that are not a GF(2) sum of strictly shorter cycles), which coincides with
 * RDKit's symmetrised SSSR on the ring systems of organic molecules (grappa_amd/featurize.py states the same definition). */
int grappa_ring_encoding(int n_atoms, int n_bonds, const int32_t* bonds, float* enc);

/* Index plan of a (batched) molecular graph -- what grappa_amd/batch.py BatchPlan holds for the kernels: CSR by destination with
 * neighbours ascending (indptr [N+1], indices [E]), the slot of every edge's reverse edge (rev [E]), per tuple level l = bond, angle,
 * proper, improper (arity 2, 3, 4, 4; idx[l] = [T[l]][arity] atom indices) the inverse incidence atom -> token rows pos * T + t
 * (inv_ptr[l] [N+1], inv_rows[l] [arity * T[l]]), and the packed incidence of the force kernel (inc_ptr [N+1], inc_code: (t << 4) |
 * (l << 2) | pos).  status_detail (optional): 1 = a bond is stored in one direction only, 2 = an atom without bonds. */
int grappa_plan_build(int N, int64_t E, const int64_t* src, const int64_t* dst, const int32_t* T, const int32_t* const* idx,
                      int32_t* indptr, int32_t* indices, int32_t* rev, int32_t* const* inv_ptr, int32_t* const* inv_rows,
                      int32_t* inc_ptr, int32_t* inc_code, int32_t* max_degree, int32_t* status_detail);

/* Connected components of the bond graph (directed edge list, both directions or one): label[a] = smallest atom index of a's component.
 * The water guard of Grappa.predict (reference utils/dgl_utils.py:210-236) walks them. */
int grappa_components(int N, int64_t E, const int64_t* src, const int64_t* dst, int32_t* label);

/* Index tables of the (atom, position) formulation of a writer's first layer (reference models/interaction_parameters.py:155-180 run once
 * per (atom, position) instead of once per token; grappa_amd/batch.py _position_tables), s = arity of the level (2..4), idx = T x s atom indices.
 * One flat array: idx_id (N x s) | invid_ptr (N + 1) | invid_rows (N s) | idx_tab (T x s) | invtab_ptr (s N + 1) | invtab_rows (s T), each
 * part at the next multiple of 4 elements.  Returns the elements written (out == NULL: needed), < 0 on error. */
long long grappa_position_tables(int N, int T, int s, const int32_t* idx, int32_t* out, long long out_len);

#ifdef __cplusplus
}
#endif
#endif
