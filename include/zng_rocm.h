/* zng_rocm.h -- C ABI of the MI355X (gfx950) backend for zlib-ng's functable hot path.
 *
 * This is the drop-in boundary: a plain C shared library (libzng_rocm.so, built
 * by hipcc from zlib-ng_amd/csrc) whose entry points are what zlib-ng's own
 * dispatch boundary -- `struct functable_s`, functable.h:26-42 -- would bind for
 * an `arch/rocm` backend.  Pointers and sizes only; no C++/torch types.
 * INTEGRATION.md shows the reference-side stub (functable.c / arch/rocm) that
 * calls these.
 *
 * Two families of entry points:
 *
 *  1. `zng_rocm_<slot>`: the functable slot itself, same argument meaning as the
 *     reference slot, HOST pointers.  Data is staged to the device through a
 *     bounded (16 MiB) staging chunk, the HIP kernel runs, the answer comes back.
 *     They never compute on the CPU: if the device is unusable they abort() with
 *     a message (a functable slot has no error channel).  Each has a twin
 *     `zng_rocm_<slot>_try` that returns a ZNG_ROCM_E* code instead of aborting,
 *     which is what the reference-side adapter binds so that it can fall back to
 *     zlib-ng's own CPU tier ("any HIP failure must degrade to the
 *     CPU implementation").
 *
 *  2. `zng_rocm_<slot>_dev`: the same operation on data ALREADY RESIDENT IN HBM
 *     (device pointers), asynchronous on a caller-supplied HIP stream
 *     (`void *stream` is a hipStream_t; NULL = the default stream).  Results are
 *     written to device memory.  This is the measured hot path (bench.py) and
 *     what a stream-level offload (DEFLATE_HOOK / INFLATE_TYPEDO_HOOK,
 *     deflate.c:72-106, inflate_p.h:11-41) calls with its device-resident window.
 *
 * All functions return 0 on success or a negative ZNG_ROCM_E* code unless they
 * mirror a reference signature that returns a value.
 */
#ifndef ZNG_ROCM_H
#define ZNG_ROCM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZNG_ROCM_OK          0
#define ZNG_ROCM_ENODEV     (-1)   /* no usable gfx950 device / HIP runtime error at init */
#define ZNG_ROCM_EHIP       (-2)   /* a HIP call failed; see zng_rocm_last_error() */
#define ZNG_ROCM_EINVAL     (-3)   /* bad argument */
#define ZNG_ROCM_ENOMEM     (-4)

/* ---- lifecycle --------------------------------------------------------- */
/* Replaces: the feature probe an arch backend adds to cpu_features
 * (cpu_features.h:23-37, x86_features.c:69-117).  Idempotent, thread-safe.
 * Builds the constant tables in HBM for `device` (-1 = current device). */
int         zng_rocm_init(int device);
int         zng_rocm_available(void);          /* 1 after a successful init */
int         zng_rocm_device_count(void);       /* 0 when there is no GPU; never aborts */
const char *zng_rocm_last_error(void);
/* {CUs, LDS bytes per CU, wavefront size, XCDs} of the initialised device */
int         zng_rocm_device_info(int32_t out[4]);
int         zng_rocm_shutdown(void);
/* Per-stream state (partials, scratch of the stream-level entry points, staging) is created lazily for every HIP
 * stream a *_dev call is given and kept until zng_rocm_shutdown().  A caller that destroys one of its streams
 * releases that state first with this call (it synchronises the stream).  The host-pointer slots use one stream
 * per host thread and release it themselves when the thread ends. */
int         zng_rocm_stream_release(void *stream);
/* The checksum kernels run one workgroup per CU for the whole pass.  A caller that overlaps them with work on
 * other HIP streams of the same device (an RCCL collective, the combine of the previous step) asks for `n` CUs
 * to be left out of that grid, so the other stream's kernels do not have to displace a workgroup the whole pass
 * waits for.  0 (the default) = use every CU.  Process-wide; takes effect at the next launch. */
int         zng_rocm_reserve_cus(int n);

/* ---- functable slots, host pointers (functable.h:26-42) ---------------- */
/* slot `adler32`: arch/generic/adler32_c.c:11-54 */
uint32_t zng_rocm_adler32(uint32_t adler, const uint8_t *buf, size_t len);
/* slot `adler32_fold_copy`: arch/generic/adler32_fold_c.c:11-15 */
uint32_t zng_rocm_adler32_fold_copy(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len);
/* slot `crc32`: arch/generic/crc32_braid_c.c:62-216 */
uint32_t zng_rocm_crc32(uint32_t crc, const uint8_t *buf, size_t len);

/* slots `crc32_fold_reset/_fold/_fold_copy/_fold_final`:
 * arch/generic/crc32_fold_c.c:10-31 over struct crc32_fold_s (crc32.h:8-14). */
typedef struct zng_rocm_crc32_fold_s {
    uint8_t  fold[64];
    uint32_t value;
} zng_rocm_crc32_fold_t;
uint32_t zng_rocm_crc32_fold_reset(zng_rocm_crc32_fold_t *crc);
void     zng_rocm_crc32_fold(zng_rocm_crc32_fold_t *crc, const uint8_t *src, size_t len, uint32_t init_crc);
void     zng_rocm_crc32_fold_copy(zng_rocm_crc32_fold_t *crc, uint8_t *dst, const uint8_t *src, size_t len);
uint32_t zng_rocm_crc32_fold_final(zng_rocm_crc32_fold_t *crc);

/* The same slots with an error channel (no reference counterpart: a functable slot returns a value only).
 * 0 and the value through *out / the updated fold state, or a negative ZNG_ROCM_E* code with nothing written and
 * the fold state unchanged; never abort().  NULL buffers behave as in the slot (adler32 -> 1, crc32 -> 0). */
int zng_rocm_adler32_try(uint32_t adler, const uint8_t *buf, size_t len, uint32_t *out);
int zng_rocm_adler32_fold_copy_try(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len, uint32_t *out);
int zng_rocm_crc32_try(uint32_t crc, const uint8_t *buf, size_t len, uint32_t *out);
int zng_rocm_crc32_fold_try(zng_rocm_crc32_fold_t *crc, const uint8_t *src, size_t len, uint32_t init_crc);
int zng_rocm_crc32_fold_copy_try(zng_rocm_crc32_fold_t *crc, uint8_t *dst, const uint8_t *src, size_t len);

/* ---- checksums on device-resident data --------------------------------- */
/* d_out: device pointer to one uint32_t.  Same value as the slot on the same
 * bytes and seed (bit-exact), including len == 0. */
int zng_rocm_adler32_dev(uint32_t adler, const void *d_buf, size_t len, uint32_t *d_out, void *stream);
int zng_rocm_crc32_dev(uint32_t crc, const void *d_buf, size_t len, uint32_t *d_out, void *stream);
/* One pass over the bytes, both checksums: d_out[0] = adler32, d_out[1] = crc32.
 * (What inflate's inf_chksum / deflate's read_buf need for a gzip+zlib pair,
 * inflate.c:25-49, deflate.c:1190-1212; BASELINE.json configs[1].) */
int zng_rocm_adler32_crc32_dev(uint32_t adler, uint32_t crc, const void *d_buf, size_t len,
                               uint32_t *d_out2, void *stream);
/* fold_copy on device: checksum `len` bytes of d_src while copying them to
 * d_dst (2N bytes of traffic).  which: 1 = adler32, 2 = crc32, 3 = both
 * (d_out2[0] adler, d_out2[1] crc; unused entries untouched). */
int zng_rocm_fold_copy_dev(int which, uint32_t adler, uint32_t crc, void *d_dst, const void *d_src,
                           size_t len, uint32_t *d_out2, void *stream);

/* combine operators, computed ON DEVICE from device-resident operands
 * (adler32.c:32-54 adler32_combine_, crc32_braid_comb.c:16-18 crc32_combine_):
 * folds `count` consecutive blocks {check[i], len[i]} left to right into one
 * checksum.  d_checks/d_lens are device arrays; d_out one uint32_t. */
int zng_rocm_adler32_combine_dev(const uint32_t *d_checks, const uint64_t *d_lens, size_t count,
                                 uint32_t *d_out, void *stream);
int zng_rocm_crc32_combine_dev(const uint32_t *d_checks, const uint64_t *d_lens, size_t count,
                               uint32_t *d_out, void *stream);

/* The same fold over packed rows {adler32, crc32, len} -- the payload the multi-GPU aggregate all-gathers
 * (one 16-byte row per rank or shard): d_out2[0] = adler32, d_out2[1] = crc32 of the
 * concatenation, rows in order.  Asynchronous on `stream`, no allocation: made to be chained behind a collective. */
typedef struct zng_rocm_check_row {
    uint32_t adler, crc;
    uint64_t len;
} zng_rocm_check_row;
int zng_rocm_combine_rows_dev(const zng_rocm_check_row *d_rows, size_t count, uint32_t *d_out2, void *stream);

/* host-side scalar forms with the reference's exact semantics
 * (zng_adler32_combine adler32.c:66-68; zng_crc32_combine/_gen/_op
 * crc32_braid_comb.c:44-54) -- used by the multi-GPU aggregate. */
uint32_t zng_rocm_adler32_combine(uint32_t adler1, uint32_t adler2, int64_t len2);
uint32_t zng_rocm_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2);
uint32_t zng_rocm_crc32_combine_gen(int64_t len2);
uint32_t zng_rocm_crc32_combine_op(uint32_t crc1, uint32_t crc2, uint32_t op);

/* ---- deflate-side primitives on device-resident stream state -------------
 * The window/prev/head slabs of `deflate_state` (deflate.h:164-244, layout
 * deflate.c:202-264) live in HBM; one `zng_rocm_deflate_view` per stream is the
 * subset of fields the functable kernels read.  A
 * stream-level offload (DEFLATE_HOOK) marshals these from its deflate_state.
 * Every call below works on an ARRAY of views in device memory: one wavefront
 * per stream, all streams of the batch in one launch.  Field meaning is the
 * reference's. */
typedef struct zng_rocm_deflate_view {
    uint8_t  *window;            /* device; 2*w_size bytes + >= 258+8 readable padding (deflate.c:1341-1372) */
    uint16_t *prev;              /* device; w_size Pos entries */
    uint16_t *head;              /* device; 65536 Pos entries (HASH_SIZE, deflate.h:81-85) */
    uint32_t  w_size, w_mask;
    uint32_t  lookahead, strstart, match_start, prev_length;
    uint32_t  max_chain_length, good_match;
    int32_t   nice_match, level;
} zng_rocm_deflate_view;

/* slot `slide_hash` (arch/generic/slide_hash_c.c:15-52) for nstreams states: every head[] and
 * prev[] entry m becomes m >= w_size ? m - w_size : 0. */
int zng_rocm_slide_hash_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, void *stream);
/* slot `compare256` (arch/generic/compare256_c.c:12-47): d_len[i] = first differing byte of
 * d_base + d_off0[i] and d_base + d_off1[i], capped at 256 (both must have 256 readable bytes). */
int zng_rocm_compare256_dev(const uint8_t *d_base, const uint64_t *d_off0, const uint64_t *d_off1, size_t npairs,
                            uint32_t *d_len, void *stream);
/* `update_hash` (insert_string.c:11-13, insert_string_tpl.h:48-51) over an array of 4-byte values */
int zng_rocm_update_hash_dev(const uint32_t *d_val, size_t n, uint32_t *d_hash, void *stream);
/* `quick_insert_string` (insert_string_tpl.h:58-75): stream i inserts position d_str[i]; d_head_out[i]
 * receives the previous chain head. */
int zng_rocm_quick_insert_string_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                                     uint16_t *d_head_out, void *stream);
/* `insert_string` (insert_string_tpl.h:85-104): stream i inserts d_count[i] consecutive positions from d_str[i],
 * in order. */
int zng_rocm_insert_string_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                               const uint32_t *d_count, void *stream);
/* The rolling-hash instantiation of the same three (insert_string_roll.c:10-24: HASH_SLIDE 5, one byte read
 * at str + 2, mask 32767), which lm_init binds for level 9 (max_chain_length > 1024, deflate.c:1223-1234).
 * The running key s->ins_h (deflate.h:196) is per stream: d_ins_h[i] is read and updated.
 * update_hash_roll: d_hash[i] = ((d_h[i] << 5) ^ (uint8_t)d_val[i]) & 32767. */
int zng_rocm_update_hash_roll_dev(const uint32_t *d_h, const uint32_t *d_val, size_t n, uint32_t *d_hash, void *stream);
int zng_rocm_quick_insert_string_roll_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                                          uint32_t *d_ins_h, uint16_t *d_head_out, void *stream);
int zng_rocm_insert_string_roll_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                                    const uint32_t *d_count, uint32_t *d_ins_h, void *stream);
/* slot `longest_match` (match_tpl.h:26-280, non-SLOW): for stream i walks the chain from d_cur_match[i];
 * d_len_out[i] = returned length, d_match_start_out[i] = s->match_start afterwards (unchanged if no
 * longer match was found). */
int zng_rocm_longest_match_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint16_t *d_cur_match,
                               uint32_t *d_len_out, uint32_t *d_match_start_out, void *stream);

/* slot `longest_match_slow` (match_tpl.h:26-280 with LONGEST_MATCH_SLOW; levels 7-9).  s->update_hash is the
 * multiplicative hash of insert_string.c:11-13 for levels 7-8 and the rolling one for level 9, chosen as
 * lm_init does by max_chain_length > 1024.  Same interface as zng_rocm_longest_match_dev. */
int zng_rocm_longest_match_slow_dev(const zng_rocm_deflate_view *d_views, size_t nstreams,
                                    const uint16_t *d_cur_match, uint32_t *d_len_out, uint32_t *d_match_start_out,
                                    void *stream);

/* Many messages in one pass -- the many-stream form of adler32 / crc32 (every block or stream of a pigz-style job has
 * its own check value, combined afterwards with zng_rocm_*_combine_dev): one workgroup per message up to 16 MiB, more
 * above, one set of launches for all of them.  which: 1 = Adler-32, 2 = CRC-32, 3 = both; per job the device buffer, its
 * length (< 16 GiB) and the seeds (adler32.c:11 / crc32.c semantics: 1 and 0 start a new check).  d_out2 receives two
 * words per job {adler, crc}; the word not asked for is left untouched.  Asynchronous on `stream`; jobs is a host array. */
typedef struct zng_rocm_check_job {
    const void *buf;      /* device */
    uint64_t    len;
    uint32_t    adler;    /* seed */
    uint32_t    crc;      /* seed */
} zng_rocm_check_job;
int zng_rocm_checksums_dev(int which, const zng_rocm_check_job *jobs, size_t njobs, uint32_t *d_out2, void *stream);

/* ---- inflate-side copy primitive ------------------------------------------
 * slot `chunkmemset_safe` (chunkset_tpl.h:229-261) as a batch of INDEPENDENT copies inside one device
 * buffer: copy i writes out = d_base + d_out_off[i], reads from = d_base + d_from_off[i], with the
 * reference's contract on [out, out + min(len,left)): forward byte-serial copy (a distance shorter than
 * the length replicates the pattern; `from` ahead of `out` behaves like memmove).  Nothing outside that
 * range is written.  Copies of one batch must not depend on each other. */
int zng_rocm_chunkmemset_safe_dev(uint8_t *d_base, const uint64_t *d_out_off, const uint64_t *d_from_off,
                                  const uint32_t *d_len, const uint32_t *d_left, size_t ncopies, void *stream);
/* slot `chunksize` (chunkset_tpl.h:9-11): store granule of the device copy kernels (bytes). */
uint32_t zng_rocm_chunksize(void);

/* ---- whole-stream deflate on device, many independent streams (level-1 class) -------------
 * The caller this replaces is deflate_quick (deflate_quick.c:47-130) behind DEFLATE_HOOK
 * (deflate.c:1039): per job ONE static-Huffman block of raw RFC 1951, built from the same primitives
 * (insert_string hash, single chain-head probe, compare256, static trees) run wavefront-wide with the
 * head table in LDS and the bit assembly in LDS (one kernel; nothing but input and output touches HBM).
 * Output is valid deflate that any inflater restores to the input; it is not bit-identical to the
 * reference's stream.
 *   in:  device; any alignment; `dict_len` (<= 32768) bytes of history sit directly in front of it
 *        (in - dict_len .. in): they prime the hash as deflateSetDictionary does (deflate.c:456-531) and
 *        matches may reach back into them, but they are neither emitted nor part of the checksum
 *   out: device, 4-byte aligned, out_cap >= zng_rocm_deflate_quick_bound(in_len)
 *   flags: 0 = the block is the stream's last (BFINAL = 1: what one deflate(Z_FINISH) call emits);
 *        ZNG_ROCM_BLOCK_NOT_FINAL = BFINAL 0; with ZNG_ROCM_BLOCK_SYNC_FLUSH an empty stored block follows
 *        (00 00 ff ff after the padding bits: the Z_SYNC_FLUSH marker, deflate.c:1064-1076), so the job's
 *        output is a whole number of bytes and jobs CONCATENATE into one valid stream -- the pigz scheme:
 *        blocks of one input compressed independently, each primed with the 32 KiB before it.
 * `jobs` is a HOST array (copied internally).  d_results (device) receives per job
 * {compressed length, adler32(1, in, in_len)} -- the {clen, check} row of the multi-stream table. */
#define ZNG_ROCM_BLOCK_NOT_FINAL  1u
#define ZNG_ROCM_BLOCK_SYNC_FLUSH 2u
typedef struct zng_rocm_stream_job {
    const uint8_t *in;
    uint8_t       *out;
    uint32_t       in_len;
    uint32_t       out_cap;
    uint32_t       dict_len;
    uint32_t       flags;
} zng_rocm_stream_job;
size_t zng_rocm_deflate_quick_bound(size_t source_len);
int    zng_rocm_deflate_quick_dev(const zng_rocm_stream_job *jobs, size_t njobs, uint32_t *d_results, void *stream);

/* ---- whole-stream deflate on device, ONE large stream (chain-walking levels) -----------------
 * The caller this replaces is deflate_medium (deflate_medium.c:145-277) + zng_tr_flush_block's
 * dynamic-tree block (trees.c:625-741) behind DEFLATE_HOOK.  The plaintext is device resident and
 * complete, so it is compressed as parallel 128-512 KiB segments (hash primed with the preceding 32 KiB,
 * so matches cross segment borders) into one continuous raw RFC 1951 stream: per segment one block whose
 * type is chosen as zng_tr_flush_block does (stored / static / dynamic, trees.c:660-719), followed by an
 * empty stored block where byte alignment needs it (as Z_SYNC_FLUSH), and a final empty static block.
 * `level` 2..9 selects max_chain_length and good_match as deflate.c:142-168 does (chain capped at 256); level 1 is
 * deflate_quick's matcher (one probe of the chain head, deflate_quick.c:89-97) on the same segment scheme; level 0
 * is deflate_stored (deflate_stored.c:27-186): stored blocks of MAX_STORED = 65535 bytes, the last one final -- the
 * split that function makes when input and output are both complete (:46-95).  d_out needs
 * zng_rocm_deflate_bound(in_len) bytes.
 * Synchronises `stream` (the segment lengths are prefix-summed on the host).  Returns 0, a ZNG_ROCM_E* code, or
 * -5 (Z_BUF_ERROR).
 * zng_rocm_deflate_block_dev is the same for one BLOCK of a longer stream: `dict_len` (<= 32768) bytes of history
 * in front of d_in prime the first segment, and `flags` (ZNG_ROCM_BLOCK_*, above) say whether the stream ends
 * here; a non-final block always ends byte aligned behind an empty stored block. */
size_t zng_rocm_deflate_bound(size_t source_len);
int    zng_rocm_deflate_dev(int level, const uint8_t *d_in, size_t in_len, uint8_t *d_out, size_t out_cap,
                            size_t *out_len, void *stream);
/* as zng_rocm_deflate_block_dev for levels 1..9, without any synchronisation: every step (matcher, block emitter, the scan that
 * places the segments, the packing) is enqueued on `stream`; d_result (device, 2 x uint64) = {compressed size, 1 if it
 * did not fit out_cap} once the stream has got there */
int    zng_rocm_deflate_async_dev(int level, const uint8_t *d_in, size_t in_len, uint32_t dict_len, uint32_t flags,
                                  uint8_t *d_out, size_t out_cap, uint64_t *d_result, void *stream);
int    zng_rocm_deflate_block_dev(int level, const uint8_t *d_in, size_t in_len, uint32_t dict_len, uint32_t flags,
                                  uint8_t *d_out, size_t out_cap, size_t *out_len, void *stream);
/* Many independent streams (or independent blocks of one input) at one of the chain levels 1..9 -- the reference's
 * many-stream model (test/pigz/CMakeLists.txt:123-200) at pigz's default level: the segments of ALL streams go through
 * one set of launches per ~1 GiB of plaintext.  Per job as zng_rocm_deflate_block_dev (dict_len bytes of history in
 * front of `in`, ZNG_ROCM_BLOCK_* flags, out_cap >= zng_rocm_deflate_bound(in_len)); synchronous; out_lens[i] (host) =
 * compressed size of stream i.  The level-1 CLASS (static Huffman, one kernel) is zng_rocm_deflate_quick_dev. */
int    zng_rocm_deflate_streams_dev(int level, const zng_rocm_stream_job *jobs, size_t njobs, size_t *out_lens,
                                    void *stream);

/* ---- inflate: host bitstream decode -> token stream -> device copy resolution -------------
 * The split of slot `inflate_fast` (inffast_tpl.h:53-318): the sequential Huffman decode loop
 * (:151-226) runs on the host and emits TOKENS instead of stores; the literal stores and match
 * copies (:155-171, :228-279 CHUNKCOPY / CHUNKMEMSET / chunkcopy_safe) are resolved on the
 * device, all segments in parallel.
 *   token (uint32): bit31 = 0 -> run of `tok` literals, taken in order from `literals`
 *                   bit31 = 1 -> match: length = ((tok >> 16) & 0xff) + 3, distance = (tok & 0xffff) + 1
 *   segs: (nsegs + 1) triples {first token, first output byte, first literal}; every segment but
 *         the last holds >= 32 KiB (and < 32 KiB + 258) of output, so a match never reaches
 *         further back than the previous segment.
 * status / msg follow zlib: 1 = Z_STREAM_END, -3 = Z_DATA_ERROR with the reference's strm->msg text
 * (inflate.c:735-917, inffast_tpl.h:189-226), -5 = input ended early, -4 = out of memory.  On an
 * error the tokens describe everything decoded before the bad symbol. */
typedef struct zng_rocm_inflate_tokens {
    uint32_t   *tokens;    size_t ntokens;
    uint8_t    *literals;  size_t nliterals;
    uint64_t   *segs;      size_t nsegs;
    uint64_t    out_len;
    size_t      in_used;
    int         status;
    const char *msg;
} zng_rocm_inflate_tokens;

int  zng_rocm_inflate_tokens_decode(const uint8_t *src, size_t src_len, zng_rocm_inflate_tokens *out);
/* the same for a stream that continues history: `window_len` (<= 32768) bytes precede its first byte -- a preset
 * dictionary (inflateSetDictionary on a raw stream, inflate.c:1214-1261) or the last bytes an earlier call produced
 * (inflate's sliding window, inflate.c:325-378) -- so a distance may reach that far beyond the start of the output
 * before it is "invalid distance too far back" (inffast_tpl.h:198-226 with whave = window_len) */
int  zng_rocm_inflate_tokens_decode_window(const uint8_t *src, size_t src_len, uint32_t window_len,
                                           zng_rocm_inflate_tokens *out);
void zng_rocm_inflate_tokens_free(zng_rocm_inflate_tokens *t);
/* Device stage on device-resident token arrays.  d_symbols: workspace of out_len uint16_t;
 * d_out: out_len bytes.  Up to four launches: per-segment resolution into 16-bit symbols (a symbol
 * >= 256 names a byte of the previous segment's last 32 KiB), the context chain over those 32 KiB
 * tails in two levels, and the final translate.  `segs` must keep the >= 32 KiB-per-segment rule
 * above (zng_rocm_inflate_tokens_decode does). */
int  zng_rocm_inflate_resolve_dev(const uint32_t *d_tokens, size_t ntokens, const uint8_t *d_literals,
                                  size_t nliterals, const uint64_t *d_segs, size_t nsegs, uint16_t *d_symbols,
                                  uint8_t *d_out, uint64_t out_len, void *stream);
/* ... and on `nthreads` host threads (<= 0: one per hardware thread, capped by the control group's CPU quota): the compressed bytes are cut into parts, each
 * thread finds a block boundary in its part (a valid dynamic block header or a sync-flush marker) and decodes from
 * there; the parts are then chained from bit 0 and joined.  Same token stream semantics, status and messages as the
 * one-thread decoder (irregular streams are simply handed to it); streams without findable boundaries (all blocks
 * fixed-Huffman or stored) decode on one thread. */
int  zng_rocm_inflate_tokens_decode_threads(const uint8_t *src, size_t src_len, uint32_t window_len, int nthreads,
                                            zng_rocm_inflate_tokens *out);
/* how many parts the calling thread's last multi-threaded decode was joined from (0 = the one-thread decoder did it) */
int  zng_rocm_inflate_threads_last_parts(void);
/* resolve with a prior window: `d_window` holds the window_len bytes that precede the stream (device);
 * d_symbols must then have room for 32768 + out_len uint16_t (the window's symbols are laid in front) */
int  zng_rocm_inflate_resolve_window_dev(const uint32_t *d_tokens, size_t ntokens, const uint8_t *d_literals,
                                         size_t nliterals, const uint64_t *d_segs, size_t nsegs, uint16_t *d_symbols,
                                         uint8_t *d_out, uint64_t out_len, const uint8_t *d_window,
                                         uint32_t window_len, void *stream);
/* One-shot raw inflate (windowBits < 0): host stream in, plaintext left in device memory at d_dst.
 * Returns the zlib status of the decode (1 = Z_STREAM_END) or a negative ZNG_ROCM_E* / Z_* code;
 * *out_len = bytes produced.  -5 also when dst_cap is too small.  A stream of 4 MiB and more is copied to the device and
 * decoded there (as zng_rocm_inflate_large_dev; zng_rocm_inflate_large_last_parts() tells); anything irregular, and every
 * smaller stream, is decoded on the calling thread (zng_rocm_inflate_tokens_decode) and resolved on the device. */
int  zng_rocm_inflate_raw(const uint8_t *src, size_t src_len, uint8_t *d_dst, size_t dst_cap, uint64_t *out_len,
                          void *stream);

/* as zng_rocm_inflate_raw, also reporting how many input bytes the deflate stream occupied */
int  zng_rocm_inflate_raw_ex(const uint8_t *src, size_t src_len, uint8_t *d_dst, size_t dst_cap, uint64_t *out_len,
                             size_t *in_used, void *stream);

/* ... and of a stream that continues `window_len` bytes of device-resident history (dictionary / sliding window) */
int  zng_rocm_inflate_raw_window(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                                 uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, void *stream);

/* ONE large raw stream that is ALREADY in device memory, inflated on the device: block starts are found on the device
 * (dynamic headers the decoder would accept, sync-flush markers, byte-aligned stored blocks), the stream is cut there into parts, one
 * wavefront decodes each part into 16-bit symbols (inffast_tpl.h:151-298 with the history still unknown), the parts are
 * chained from bit 0 and the context chain of zng_rocm_inflate_resolve_dev turns the symbols into bytes.  The host only
 * sorts a few thousand candidates and walks the chain.  Streams that offer nothing to cut at (fixed-Huffman blocks
 * only) and every irregular stream go to the sequential decoder, so status and message are always the
 * reference's.  Returns as zng_rocm_inflate_raw_window; synchronous. */
int  zng_rocm_inflate_large_dev(const uint8_t *d_src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                                uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, void *stream);
/* parts on the chain of the calling thread's last zng_rocm_inflate_large_dev or zng_rocm_inflate_raw* call (0: the
 * sequential decoder did it) */
int  zng_rocm_inflate_large_last_parts(void);

/* Many independent raw streams at once: `nthreads` host threads (<= 0: as many as the host gives us) take the jobs in
 * order, each decoding on the host and resolving on the device on its own HIP stream, so that the sequential decode
 * -- where an inflate spends its time -- runs on all the cores the caller allows while the device work of one stream
 * overlaps the decode of the next (the pigz shape: test/pigz/CMakeLists.txt).  Per job: `src` host, `d_dst` device
 * (dst_cap bytes), an optional device-resident window / dictionary in front (as zng_rocm_inflate_raw_window); on
 * return `status` is the job's zlib status (1 = Z_STREAM_END, -3 with the reference's text in `msg`, -5 input ended
 * early or destination too small, negative ZNG_ROCM_E*), `out_len` / `in_used` as in the one-shot call.  Synchronous:
 * every plaintext is in place when the call returns.  The return value is 0 or the first device error. */
typedef struct zng_rocm_inflate_job {
    const uint8_t *src;       size_t   src_len;
    uint8_t       *d_dst;     size_t   dst_cap;
    const uint8_t *d_window;  uint32_t window_len;
    int            status;
    uint64_t       out_len;
    size_t         in_used;
    const char    *msg;
} zng_rocm_inflate_job;
int  zng_rocm_inflate_many(zng_rocm_inflate_job *jobs, size_t njobs, int nthreads);

/* Many independent raw deflate streams that are ALREADY in device memory, decoded entirely on the device: one wavefront
 * per stream runs slot `inflate_fast` (inffast_tpl.h:53-318) and the block decoding around it (inflate.c:735-917,
 * inftrees.c:32-297) -- Huffman decode and copies -- with the last 4 KiB of output in an LDS ring (inflate's sliding
 * window, inflate.c:325-378).  The inverse of zng_rocm_deflate_quick_dev for the reference's many-stream model
 * (test/pigz/CMakeLists.txt:123-200): nothing crosses PCIe.  `dict_len` bytes of history (a dictionary,
 * inflateSetDictionary on a raw stream inflate.c:1214-1261, or the previous window) must sit directly in front of `out`.
 * Asynchronous on `stream`; jobs is a host array (copied before the call returns).  The compressed words are fetched as
 * aligned dwords: up to 3 bytes on either side of [in, in + in_len) inside the same 4-byte words are read (never used).
 * d_results: 4 uint32 per job: {bytes produced, input bytes consumed, status as int32 (1 = Z_STREAM_END, -3 = Z_DATA_ERROR,
 * -5 = Z_BUF_ERROR: input ended early or out_cap too small), message id for zng_rocm_inflate_message}. */
typedef struct zng_rocm_inflate_dev_job {
    const void *in;        /* device: raw deflate stream */
    void       *out;       /* device: plaintext */
    uint64_t    in_len;    /* < 2 GiB */
    uint64_t    out_cap;   /* < 2 GiB */
    uint32_t    dict_len;  /* <= 32768 */
    uint32_t    flags;     /* 0 */
} zng_rocm_inflate_dev_job;
int  zng_rocm_inflate_streams_dev(const zng_rocm_inflate_dev_job *jobs, size_t njobs, uint32_t *d_results, void *stream);
/* the reference's strm->msg text for a message id of d_results ("" for 0 / unknown ids) */
const char *zng_rocm_inflate_message(uint32_t id);

/* zlib (format 1) / gzip (format 2) / raw (format 0) framing for MANY device-resident streams -- compress2 / uncompress2
 * (compress.c:31-69, uncompr.c:25-76) in the shape of the many-stream model, every step on the device, asynchronous on
 * `stream`:
 *   compress: the level-1 class per stream + (gzip) the CRC-32 of every plaintext in one pass + one kernel that writes
 *   every header and trailer.  Both wrappers are 12 bytes long so that the block starts 4-byte aligned: gzip declares an
 *   empty FEXTRA field, zlib puts two empty stored blocks behind its 2-byte header.  Per job: out 4-byte aligned,
 *   out_cap >= zng_rocm_compress_streams_bound(in_len, format), no dictionary / flags for a wrapped stream.
 *   d_results: 2 words per job {total bytes written, check value (Adler-32; gzip: CRC-32)}.
 *   uncompress: every header parsed on the device (inflate.c:509-555, :556-700 incl. FHCRC), zng_rocm_inflate_streams_dev's
 *   kernel, the check values of all outputs in one many-message pass whose descriptors are filled on the device, every
 *   trailer compared (inflate.c:1105-1147).  d_results: 4 words per job as zng_rocm_inflate_streams_dev, bytes consumed
 *   counting header and trailer; message ids include "incorrect header check", "unknown compression method", "invalid
 *   window size", "header crc mismatch", "need dictionary", "incorrect data check", "incorrect length check". */
size_t zng_rocm_compress_streams_bound(size_t source_len, int format);
int  zng_rocm_compress_streams_dev(int format, const zng_rocm_stream_job *jobs, size_t njobs, uint32_t *d_results, void *stream);
int  zng_rocm_uncompress_streams_dev(int format, const zng_rocm_inflate_dev_job *jobs, size_t njobs, uint32_t *d_results,
                                     void *stream);

/* ONE raw stream with its host decode spread over `nthreads` threads (zng_rocm_inflate_tokens_decode_threads) and one
 * device pass; same results and status as zng_rocm_inflate_raw_window, which it falls back to for streams that
 * offer no block boundary to cut at or turn out irregular.  Synchronous. */
int  zng_rocm_inflate_raw_threads(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                                  uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, int nthreads);

/* ---- compress2 / uncompress2 class front ends (compress.c:31-98, uncompr.c:25-76) ---------------------
 * `format`: 0 = raw deflate, 1 = zlib (RFC 1950), 2 = gzip (RFC 1952).  The trailer checksum (Adler-32 /
 * CRC-32 + ISIZE) is computed by the device checksum kernel over the device-resident plaintext.
 * compress2_dev:   d_src (device) -> d_dst (device, *dst_len >= zng_rocm_compress_bound()); level as compress2:
 *                  -1 = 6, 0 = stored blocks, 1..9 as zng_rocm_deflate_dev, anything else Z_STREAM_ERROR (-2,
 *                  deflate.c:318-320); the zlib FLEVEL / gzip XFL hints are those of the requested level
 *                  (deflate.c:868-885, :913).  Returns Z_OK (0) / Z_BUF_ERROR (-5) / error.
 * uncompress2_dev: src (HOST, the sequential bitstream stays on the host) -> d_dst (device).  On return *dst_len
 *                  = plaintext bytes, *src_len = input bytes consumed.  Z_OK, Z_BUF_ERROR (destination too small),
 *                  Z_DATA_ERROR with the reference's message text in zng_rocm_last_error() ("incorrect header
 *                  check", "header crc mismatch" (gzip FHCRC, inflate.c:686-692), "incorrect data check",
 *                  "incorrect length check", decoder messages, incomplete stream). */
size_t zng_rocm_compress_bound(size_t source_len, int format);
int    zng_rocm_compress2_dev(uint8_t *d_dst, size_t *dst_len, const uint8_t *d_src, size_t src_len, int level,
                              int format, void *stream);
int    zng_rocm_uncompress2_dev(uint8_t *d_dst, size_t *dst_len, const uint8_t *src, size_t *src_len, int format,
                                void *stream);

/* ---- the coarse boundary: what DEFLATE_HOOK / INFLATE_TYPEDO_HOOK call ---------------------------------------
 * A `zng_rocm_hook` is the content of an arch/rocm backend's arch_deflate_state / arch_inflate_state
 * (deflate.h:319-321, inflate.h:160-162; precedent arch/s390/dfltcc_common.h): the stream's 32 KiB of history in
 * device memory, a bounded staging buffer, a HIP stream of its own.  Host pointers in and out -- zng_stream's next_in /
 * next_out are the caller's memory (SURVEY.md 8b "Ownership").  The reference-side adapter that keeps next_in /
 * avail_in / next_out / avail_out / total_* / adler and answers need_more / block_done / finish_started /
 * finish_done (deflate.h:336-341, deflate.c:1039-1083) is integration/arch/rocm/rocm_deflate.c, rocm_inflate.c.
 * Every function returns ZNG_ROCM_ENODEV after a zng_rocm_shutdown() (the adapter then continues in software). */
typedef struct zng_rocm_hook zng_rocm_hook;
int    zng_rocm_hook_create(zng_rocm_hook **h, size_t block_bytes);       /* ENODEV without a device */
void   zng_rocm_hook_destroy(zng_rocm_hook *h);
int    zng_rocm_hook_reset(zng_rocm_hook *h);                             /* forget the history: deflateResetKeep (deflate.c:567), Z_FULL_FLUSH (deflate.c:1073-1080), inflateResetKeep */
/* deflateSetDictionary / inflateSetDictionary (deflate.c:456-531, inflate.c:1214-1261): the last 32 KiB become the history */
int    zng_rocm_hook_set_history(zng_rocm_hook *h, const uint8_t *dict, uint32_t len);
/* deflateGetDictionary / inflateGetDictionary (deflate.c:521-545, inflate.c:1195-1212): dict may be NULL (length only) */
int    zng_rocm_hook_get_history(zng_rocm_hook *h, uint8_t *dict, uint32_t *len);
size_t zng_rocm_hook_deflate_bound(size_t in_len);
/* One block of the stream: `in` (host) is compressed at `level` (0..9, as zng_rocm_deflate_block_dev) against the
 * history, the block lands at `out` (host, out_cap >= zng_rocm_hook_deflate_bound(in_len)) and always ends on a byte
 * boundary; flags = ZNG_ROCM_BLOCK_*; the input becomes history.  check: 0 none, 1 Adler-32, 2 CRC-32 of `in`,
 * continuing *check_value (what DEFLATE_NEED_CHECKSUM = 0 leaves to the backend, deflate.c:1197-1212). */
int    zng_rocm_hook_deflate_block(zng_rocm_hook *h, int level, const uint8_t *in, size_t in_len, uint32_t flags, int check,
                                   uint32_t *check_value, uint8_t *out, size_t out_cap, size_t *out_len);
/* A complete raw deflate stream (or the rest of one) at `in`, continuing the history: returns 1 (Z_STREAM_END) with the
 * plaintext at *out (host memory owned by the hook, valid until its next call), *out_len, *in_used and the check of the
 * plaintext; -5 when the stream does not end inside in_len (nothing consumed: the adapter gathers more input);
 * -3 (Z_DATA_ERROR) with the reference's strm->msg text in *msg; negative ZNG_ROCM_E* on a device failure. */
int    zng_rocm_hook_inflate(zng_rocm_hook *h, const uint8_t *in, size_t in_len, int check, uint32_t *check_value,
                             const uint8_t **out, size_t *out_len, size_t *in_used, const char **msg);

/* ---- measurement hooks --------------------------------------------------
 * Between trace_begin and trace_end every (sampled) launch of the DOMINANT kernel of a *_dev entry point (the
 * streaming kernel, not its finalize step) carries a pair of HIP events attached to its own dispatch
 * (hipExtLaunchKernelGGL): the start / stop timestamps of the kernel itself, the quantity rocprofv3
 * --kernel-trace reports, with no additional packets in the stream.  trace_end synchronises, writes the
 * per-launch durations (ms) into `ms_out` (up to `cap`) and returns how many launches were recorded
 * (negative = error).  Used by bench.py for the roofline figure; no reference counterpart (the reference
 * measures with Google Benchmark, test/benchmarks/). */
int zng_rocm_trace_begin(int max_launches);
int zng_rocm_trace_end(float *ms_out, int cap);
/* Time only every n-th marked launch (n >= 1; 1 = all, the default): a timed loop can sample its kernel instead of
 * attaching events to every step.  Takes effect at the next zng_rocm_trace_begin(). */
int zng_rocm_trace_stride(int n);

#ifdef __cplusplus
}
#endif
#endif /* ZNG_ROCM_H */
