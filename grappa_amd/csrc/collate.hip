// Device-side collate: a batch is assembled in HBM from a dataset that is RESIDENT in HBM.
// The reference collates on the host per batch (data/GraphDataLoader.py:23-73: dgl.unbatch-style deep copies, set_number_confs,
// dgl.batch); here every per-molecule table (features, CSR, tuple tables, inverse incidences, conformations) is packed once,
// and a batch is ONE launch: workgroup (slot j, table t) copies molecule ids[j]'s rows of table t to their place in the batch
// table, shifting indices by the slot's atom / edge / tuple offsets and selecting conformations on the way.  Integer and byte
// work: bit-exact with the host path.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void collate_kernel(const grappa_collate_desc* __restrict__ descs, int B) {
    const int j = blockIdx.x;
    const grappa_collate_desc d = descs[blockIdx.y];
    const int64_t r0 = d.dst_row[j], nrows = d.dst_row[j + 1] - r0;
    const int64_t s0 = d.src_row[j];
    const int w = d.width;
    if (d.mode == GRAPPA_COLLATE_CONF) {
        // row r of the slot = one atom (or the molecule, for energies): (C_src, w) values in the source, (C_out, w) in the batch
        const int c_src = d.p0[j], c_out = (int)d.c0;
        const int32_t* sel = d.p1 + (int64_t)j * c_out;
        const int64_t per_row = (int64_t)c_out * w, total = nrows * per_row;
        for (int64_t i = threadIdx.x; i < total; i += 256) {
            const int64_t r = i / per_row, rem = i - r * per_row;
            const int c = (int)(rem / w), e = (int)(rem - (int64_t)c * w);
            d.dst[(r0 + r) * per_row + rem] = d.src[s0 + (r * c_src + sel[c]) * w + e];
        }
        return;
    }
    const int64_t total = nrows * w;
    const int32_t* src = d.src + s0 * w;
    int32_t* dst = d.dst + r0 * w;
    switch (d.mode) {
        case GRAPPA_COLLATE_COPY:
            for (int64_t i = threadIdx.x; i < total; i += 256) dst[i] = src[i];
            break;
        case GRAPPA_COLLATE_ADD: {
            const int32_t off = d.p0[j];
            for (int64_t i = threadIdx.x; i < total; i += 256) dst[i] = src[i] + off;
            break;
        }
        case GRAPPA_COLLATE_INV_ROWS: {
            // local token row = pos * T_mol + t  ->  batch token row = pos * T_batch + t_off + t
            const int32_t t_mol = d.p0[j], t_off = d.p1[j];
            const int32_t t_batch = (int32_t)d.c0;
            for (int64_t i = threadIdx.x; i < total; i += 256) {
                const int32_t v = src[i], pos = v / t_mol;
                dst[i] = pos * t_batch + t_off + (v - pos * t_mol);
            }
            break;
        }
        case GRAPPA_COLLATE_INC_CODE: {
            // code = tuple << 4 | level << 2 | pos: the tuple index moves by the slot's offset at that level
            const int32_t* t_off = d.p0 + 4 * (int64_t)j;
            for (int64_t i = threadIdx.x; i < total; i += 256) {
                const int32_t v = src[i];
                dst[i] = v + (t_off[(v >> 2) & 3] << 4);
            }
            break;
        }
        default:
            break;
    }
}

}  // namespace

extern "C" int grappa_collate_batch(void* stream, const grappa_collate_desc* descs_device, const grappa_collate_desc* descs_host, int n_tables,
                                    int B) {
    if (n_tables < 0 || B < 0 || (n_tables > 0 && (!descs_device || !descs_host))) return GRAPPA_ERR_ARG;
    if (n_tables == 0 || B == 0) return GRAPPA_OK;
    if (n_tables > 65535) return GRAPPA_ERR_ARG;
    for (int t = 0; t < n_tables; ++t) {
        const grappa_collate_desc& d = descs_host[t];
        if (!d.src || !d.dst || !d.src_row || !d.dst_row || d.width < 1) return GRAPPA_ERR_ARG;
        if (d.mode < GRAPPA_COLLATE_COPY || d.mode > GRAPPA_COLLATE_CONF) return GRAPPA_ERR_ARG;
        if (d.mode != GRAPPA_COLLATE_COPY && !d.p0) return GRAPPA_ERR_ARG;
        if ((d.mode == GRAPPA_COLLATE_INV_ROWS || d.mode == GRAPPA_COLLATE_CONF) && !d.p1) return GRAPPA_ERR_ARG;
        if (d.mode == GRAPPA_COLLATE_INV_ROWS && d.width != 1) return GRAPPA_ERR_ARG;
    }
    GRAPPA_LAUNCH(collate_kernel, dim3(B, n_tables), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), descs_device, B);
    return grappa_launch_status();
}
