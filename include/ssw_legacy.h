/*
 * ssw_legacy.h -- the six symbols of the reference's libssw.so, exported by libclh.so with the same
 * signatures, struct layout and ownership rules, computed on the GPU.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the CIRI-long tree):
 *
 *   ssw_init          libs/striped_smith_waterman/ssw.h:72    (ssw.c:750-771)   bound at ssw_wrap.py:58-60
 *   init_destroy      libs/striped_smith_waterman/ssw.h:77    (ssw.c:773-777)   bound at ssw_wrap.py:62-64
 *   ssw_align         libs/striped_smith_waterman/ssw.h:112-120 (ssw.c:779-869) bound at ssw_wrap.py:66-68
 *   align_destroy     libs/striped_smith_waterman/ssw.h:125   (ssw.c:871-874)   bound at ssw_wrap.py:70-72
 *   cigar_int_to_op   libs/striped_smith_waterman/ssw.h:176   (ssw.c:876-896)   bound at ssw_wrap.py:286-288
 *   cigar_int_to_len  libs/striped_smith_waterman/ssw.h:182   (ssw.c:898-902)   bound at ssw_wrap.py:282-284
 *
 * Conventions kept: the profile BORROWS `read` and `mat` (ssw.c:766-767); s_align and its cigar are malloc'd by
 * the library and released by align_destroy; errors return NULL with a message on stderr (ssw.c:810-813,818-821,
 * 857-860); maskLen < 15 prints the reference's warning and zeroes score2 (ssw.c:799-801,826-832).
 * Differences: one extra failure mode (no usable GPU -> NULL + message) and gap_open < gap_extend is rejected.
 * The device is chosen by the environment variable CIRI_LONG_DEVICE (default 0).
 */
#ifndef CLH_SSW_LEGACY_H
#define CLH_SSW_LEGACY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)     /* the library itself is built with -fvisibility=hidden: only the C ABI is exported */

struct _profile;
typedef struct _profile s_profile;

typedef struct {            /* ssw.h:42-52, mirrored by ssw_wrap.py:29-37 */
    uint16_t score1;
    uint16_t score2;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    int32_t ref_end2;
    uint32_t* cigar;
    int32_t cigarLen;
} s_align;

s_profile* ssw_init(const int8_t* read, const int32_t readLen, const int8_t* mat, const int32_t n, const int8_t score_size);
void init_destroy(s_profile* p);
s_align* ssw_align(const s_profile* prof, const int8_t* ref, int32_t refLen, const uint8_t weight_gapO,
                   const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters, const int32_t filterd,
                   const int32_t maskLen);
void align_destroy(s_align* a);
char cigar_int_to_op(uint32_t cigar_int);
uint32_t cigar_int_to_len(uint32_t cigar_int);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
