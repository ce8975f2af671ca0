/* arch/rocm/rocm_deflate.h -- the hook block deflate.c includes instead of its defaults when ROCM_DEFLATE is defined
 * (deflate.c:72-106; precedent arch/s390/dfltcc_deflate.h:17-56).  One addition the reference does not have yet:
 * DEFLATE_END_HOOK, invoked from deflateEnd() before free_deflate() -- DFLTCC owns no resources, a GPU backend does.
 * Function names carry `archrocm_`, not `rocm_`: under the native API PREFIX() prepends `zng_`, and `zng_rocm_*` is the name space of
 * libzng_rocm's own entry points (a `zng_rocm_deflate_bound` here would shadow the library's).
 */
#ifndef ROCM_DEFLATE_H_
#define ROCM_DEFLATE_H_
#include "rocm_common.h"

void Z_INTERNAL PREFIX(archrocm_reset_deflate_state)(PREFIX3(streamp) strm);
void Z_INTERNAL PREFIX(archrocm_deflate_end)(PREFIX3(streamp) strm);
int  Z_INTERNAL PREFIX(archrocm_can_deflate)(PREFIX3(streamp) strm);
int  Z_INTERNAL PREFIX(archrocm_deflate)(PREFIX3(streamp) strm, int flush, block_state *result);
int  Z_INTERNAL PREFIX(archrocm_deflate_params)(PREFIX3(streamp) strm, int level, int strategy, int *flush);
int  Z_INTERNAL PREFIX(archrocm_deflate_done)(PREFIX3(streamp) strm, int flush);
int  Z_INTERNAL PREFIX(archrocm_deflate_set_dictionary)(PREFIX3(streamp) strm, const unsigned char *dictionary, unsigned dict_length);
int  Z_INTERNAL PREFIX(archrocm_deflate_get_dictionary)(PREFIX3(streamp) strm, unsigned char *dictionary, unsigned *dict_length);
size_t Z_INTERNAL PREFIX(archrocm_deflate_bound)(size_t source_len);

#define DEFLATE_SET_DICTIONARY_HOOK(strm, dict, dict_len) \
    do { \
        if (PREFIX(archrocm_can_deflate)((strm))) \
            return PREFIX(archrocm_deflate_set_dictionary)((strm), (dict), (dict_len)); \
    } while (0)
#define DEFLATE_GET_DICTIONARY_HOOK(strm, dict, dict_len) \
    do { \
        if (PREFIX(archrocm_can_deflate)((strm))) \
            return PREFIX(archrocm_deflate_get_dictionary)((strm), (dict), (dict_len)); \
    } while (0)
#define DEFLATE_RESET_KEEP_HOOK PREFIX(archrocm_reset_deflate_state)
#define DEFLATE_END_HOOK PREFIX(archrocm_deflate_end)
#define DEFLATE_PARAMS_HOOK(strm, level, strategy, hook_flush) \
    do { \
        int err = PREFIX(archrocm_deflate_params)((strm), (level), (strategy), (hook_flush)); \
        if (err == Z_STREAM_ERROR) \
            return err; \
    } while (0)
#define DEFLATE_DONE PREFIX(archrocm_deflate_done)
#define DEFLATE_BOUND_ADJUST_COMPLEN(strm, complen, source_len) \
    do { \
        if (deflateStateCheck((strm)) || PREFIX(archrocm_can_deflate)((strm))) \
            (complen) = PREFIX(archrocm_deflate_bound)(source_len); \
    } while (0)
#define DEFLATE_NEED_CONSERVATIVE_BOUND(strm) (PREFIX(archrocm_can_deflate)((strm)))
#define DEFLATE_HOOK PREFIX(archrocm_deflate)
#define DEFLATE_NEED_CHECKSUM(strm) (!PREFIX(archrocm_can_deflate)((strm)))
/* the device's output is a function of the input and of where the caller flushes, nothing else */
#define DEFLATE_CAN_SET_REPRODUCIBLE(strm, reproducible) 1
#define DEFLATE_ADJUST_WINDOW_SIZE(n) (n)
#endif
