/* arch/rocm/rocm_inflate.h -- the hook block inflate_p.h includes instead of its defaults when ROCM_INFLATE is defined
 * (inflate_p.h:11-41; precedent arch/s390/dfltcc_inflate.h:21-65).  One addition the reference does not have yet:
 * INFLATE_END_HOOK, invoked from inflateEnd() before free_inflate(). */
#ifndef ROCM_INFLATE_H_
#define ROCM_INFLATE_H_
#include "rocm_common.h"

typedef enum {
    ROCM_INFLATE_CONTINUE,      /* state->mode has been set: go round inflate()'s loop again */
    ROCM_INFLATE_BREAK,         /* leave inflate() with *ret */
    ROCM_INFLATE_SOFTWARE       /* not ours: decode the block in software */
} rocm_inflate_action;

void Z_INTERNAL PREFIX(archrocm_reset_inflate_state)(PREFIX3(streamp) strm);
void Z_INTERNAL PREFIX(archrocm_inflate_end)(PREFIX3(streamp) strm);
int  Z_INTERNAL PREFIX(archrocm_can_inflate)(PREFIX3(streamp) strm);
rocm_inflate_action Z_INTERNAL PREFIX(archrocm_inflate)(PREFIX3(streamp) strm, int flush, int *ret);
int  Z_INTERNAL PREFIX(archrocm_was_inflate_used)(PREFIX3(streamp) strm);
int  Z_INTERNAL PREFIX(archrocm_inflate_disable)(PREFIX3(streamp) strm);
int  Z_INTERNAL PREFIX(archrocm_inflate_set_dictionary)(PREFIX3(streamp) strm, const unsigned char *dictionary, unsigned dict_length);
int  Z_INTERNAL PREFIX(archrocm_inflate_get_dictionary)(PREFIX3(streamp) strm, unsigned char *dictionary, unsigned *dict_length);

#define INFLATE_RESET_KEEP_HOOK PREFIX(archrocm_reset_inflate_state)
#define INFLATE_END_HOOK PREFIX(archrocm_inflate_end)
#define INFLATE_PRIME_HOOK(strm, bits, value) \
    do { if (PREFIX(archrocm_inflate_disable)((strm))) return Z_STREAM_ERROR; } while (0)
#define INFLATE_TYPEDO_HOOK(strm, flush) \
    if (PREFIX(archrocm_can_inflate)((strm))) { \
        rocm_inflate_action action; \
\
        RESTORE(); \
        action = PREFIX(archrocm_inflate)((strm), (flush), &ret); \
        LOAD(); \
        if (action == ROCM_INFLATE_CONTINUE) \
            break; \
        else if (action == ROCM_INFLATE_BREAK) \
            goto inf_leave; \
    }
#define INFLATE_NEED_CHECKSUM(strm) (!PREFIX(archrocm_can_inflate)((strm)))
#define INFLATE_NEED_UPDATEWINDOW(strm) (!PREFIX(archrocm_can_inflate)((strm)))
#define INFLATE_MARK_HOOK(strm) \
    do { if (PREFIX(archrocm_was_inflate_used)((strm))) return -(1L << 16); } while (0)
#define INFLATE_SYNC_POINT_HOOK(strm) \
    do { if (PREFIX(archrocm_was_inflate_used)((strm))) return Z_STREAM_ERROR; } while (0)
#define INFLATE_SET_DICTIONARY_HOOK(strm, dict, dict_len) \
    do { \
        if (PREFIX(archrocm_can_inflate)((strm))) \
            return PREFIX(archrocm_inflate_set_dictionary)((strm), (dict), (dict_len)); \
    } while (0)
#define INFLATE_GET_DICTIONARY_HOOK(strm, dict, dict_len) \
    do { \
        if (PREFIX(archrocm_can_inflate)((strm))) \
            return PREFIX(archrocm_inflate_get_dictionary)((strm), (dict), (dict_len)); \
    } while (0)
#define INFLATE_ADJUST_WINDOW_SIZE(n) (n)
#endif
