/* chunkset_oracle.c -- CPU restatement of the inflate match-copy primitive.
 * TEST INFRASTRUCTURE ONLY (see zng_oracle.h).
 *
 * Follows /root/reference (zlib-ng 2.2.2):
 *   chunkset_tpl.h:9-11      CHUNKSIZE          -> oracle_chunksize (generic chunk_t = uint64_t,
 *                                                  arch/generic/chunkset_c.c:7-9)
 *   chunkset_tpl.h:229-261   CHUNKMEMSET_SAFE   -> oracle_chunkmemset_safe
 *   chunkset_tpl.h:112-227   CHUNKMEMSET (dist==1 memset, dist>=chunk CHUNKCOPY :24-41,
 *                            short-distance magazine GET_CHUNK_MAG :69-87, from-ahead memmove :125-140)
 *
 * Every branch of the reference computes, on the caller-visible range
 * [out, out + MIN(len,left)), the forward byte-serial copy out[i] = from[i]
 * (i ascending; so dist < len replicates the dist-byte pattern and a `from`
 * ahead of `out` behaves like memmove).  Bytes in [out+len, out+left) may be
 * scribbled by the reference's chunk stores and are NOT part of the contract
 * (SURVEY.md section 9.1), so the oracle leaves them alone and tests compare
 * only the contractual range.
 */
#include "zng_oracle.h"

uint32_t oracle_chunksize(void) {
    return 8;   /* sizeof(uint64_t), arch/generic/chunkset_c.c:7 */
}

uint8_t *oracle_chunkmemset_safe(uint8_t *out, uint8_t *from, unsigned len, unsigned left) {
    if (len > left)             /* chunkset_tpl.h:236 */
        len = left;
    for (unsigned i = 0; i < len; i++)
        out[i] = from[i];
    return out + len;
}
