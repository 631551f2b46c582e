// plat_assemble.hip -- coloured de-Bruijn local assembler.
#include "plat_internal.hpp"

PLAT_EXPORT int plat_assemble_batch(plat_ctx* ctx, const plat_assembly_batch* batch, int kmer_size, int min_qual,
                                    int min_weight, int no_cycles, int max_vars_per_region, int blob_per_region,
                                    int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd,
                                    int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream)
{
    (void)batch; (void)kmer_size; (void)min_qual; (void)min_weight; (void)no_cycles; (void)max_vars_per_region;
    (void)blob_per_region; (void)var_count; (void)var_pos; (void)var_nrem; (void)var_nadd; (void)var_off;
    (void)var_blob; (void)status; (void)stream;
    if (!ctx) return PLAT_ERR_INVALID;
    return PLAT_ERR_UNSUPPORTED;   // device assembler not built yet: fail loudly, never fall back to the CPU
}
