/* pfa.h — C ABI of libpfa_hip.so: the MI355X (gfx950) polymorphism-scan path of
 * PolyFastA, behind plain pointers and sizes (no C++ or torch types).
 *
 * What this boundary replaces.  The reference (PolyFastA.py, pure Python) has no
 * FFI; its only seam is the Python call boundary around the statistics kernel,
 * used from one place, print_result.no_header (PolyFastA.py:148-198):
 *
 *     getvarsites(d, seqlen)      -> (pos, var)                PolyFastA.py:252-261
 *     getsfs(var)                 -> folded SFS                PolyFastA.py:274-282
 *     getvarCDSsites(d, seqlen)   -> (syn sites, S, N, nstops, missing)   :284-315
 *     get_syn_nonsyn_cod_sites()  -> codon-set classifier      PolyFastA.py:319-434
 *     polymorphism(var, len, jc)  -> (S, pi, theta_w, D)       PolyFastA.py:502-520
 *
 * A maintainer binds this library with ctypes (INTEGRATION.md shows the stub)
 * and replaces those calls: the device returns exact INTEGER summaries per scan
 * unit (an alignment, or a population's row subset of one); every
 * floating-point step (pi, theta_w, Tajima's D, Jukes-Cantor, "NA" rules, CSV
 * text) stays in the host language, where its rounding mirrors the reference.
 *
 * Conventions: every function returns 0 on success or a negative PFA_ERR_*
 * code and records a message retrievable with pfa_last_error() (thread-local).
 * No exceptions or aborts cross the ABI.  One process drives one GPU
 * (pfa_init(device)); calls are not re-entrant across threads.  The caller owns
 * every host buffer; the library keeps no host pointer after a call returns.
 * There is NO CPU fallback: without a gfx950 device pfa_init() fails.
 */
#ifndef PFA_H
#define PFA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFA_ABI_VERSION 7

/* error codes */
#define PFA_OK 0
#define PFA_ERR_NO_DEVICE (-1)  /* no HIP device / wrong architecture        */
#define PFA_ERR_NOT_INIT (-2)   /* pfa_init() has not succeeded               */
#define PFA_ERR_ARG (-3)        /* bad argument or descriptor                 */
#define PFA_ERR_HIP (-4)        /* a HIP runtime call failed                  */
#define PFA_ERR_TOO_MANY_ROWS (-5) /* n_rows above PFA_MAX_ROWS               */
#define PFA_ERR_NOMEM (-6)
#define PFA_ERR_FORMAT (-7)     /* FASTA packer: not FASTA / unsupported byte */
#define PFA_ERR_IO (-8)
#define PFA_ERR_UNSUPPORTED (-9) /* FASTA packer: a byte >= 0x80 in a sequence line */

/* Rows per scan unit.  The reference has no limit (PolyFastA.py:252-261, :284-315); here every count is 32 bits wide and every
 * product of counts 64 (units with many rows are streamed in row chunks; coding units above 2 048 rows keep their per-site H in 56
 * bits), so the bound is where the tests end: 2^24 rows in one unit (a 0.27 GB matrix at 16 sites; checked at that size against the
 * C oracle and at 2^21 + 3 rows against the reference itself, tests/golden G10).  One row more is refused with
 * PFA_ERR_TOO_MANY_ROWS, never answered wrongly. */
#define PFA_MAX_ROWS 16777216u
#define PFA_MAX_ROWS_CDS PFA_MAX_ROWS

#define PFA_NO_ROW_IDX 0xFFFFFFFFu

/* pfa_desc.flags */
#define PFA_FLAG_CDS 1u /* in-frame coding sequence: also produce the syn/nonsyn split */

/* pfa_plan_create() `want` bits: optional per-site outputs */
#define PFA_WANT_VAR_BITMAP 1u /* variable-site bitmap  -> `pos` of getvarsites        */
#define PFA_WANT_SN_BITMAP 2u  /* S / N position bitmaps -> lists of getvarCDSsites    */

/* Bitmaps and pfa_desc.tile_off are laid out per 256-site tile (whatever tile
 * width the kernel that serves a unit works on; pfa_plan_describe()). */
#define PFA_TILE_SITES 256u

/* One scan unit = one alignment, or one population's row subset of an
 * alignment (PolyFastA.py:122-134).  `bases` holds row-major matrices, ONE
 * UPPER-CASE ASCII BYTE PER BASE exactly as readfasta (PolyFastA.py:227-250)
 * would produce them; every distinct byte value is an allele (PolyFastA.py:258). */
typedef struct pfa_desc {
  uint64_t base_off;    /* byte offset of (row 0, site 0) in `bases`; multiple of 16        */
  uint32_t n_rows;      /* N: sequences in this unit (<= PFA_MAX_ROWS)                       */
  uint32_t n_sites;     /* L: alignment length                                               */
  uint32_t row_stride;  /* bytes between matrix rows; multiple of 16 and >= round_up(L, 16)  */
  uint32_t row_idx_off; /* offset into row_idx[] of the unit's n_rows matrix row numbers,
                           or PFA_NO_ROW_IDX for rows 0..n_rows-1                            */
  uint32_t flags;       /* PFA_FLAG_*                                                        */
  uint32_t sfs_off;     /* offset (in uint32) of this unit's floor(N/2) SFS bins in `sfs`    */
  uint32_t tile_off;    /* index of this unit's first tile in the bitmap outputs             */
  uint32_t mat_rows;    /* rows of the underlying matrix (bounds check of row_idx entries)   */
} pfa_desc;

/* Integer summary of one scan unit.  theta_pi total =
 * H / (N (N-1)); theta_w total = S / a1; see polyfasta_amd/stats.py. */
typedef struct pfa_result {
  uint64_t H;            /* sum over variable sites of (N^2 - sum_j c_j^2)   (PolyFastA.py:485-492) */
  uint64_t H_S;          /* same, over sites called synonymous               (PolyFastA.py:169,172) */
  uint64_t H_N;          /* same, over sites called non-synonymous           (PolyFastA.py:170,173) */
  uint32_t seg_sites;    /* S = len(var)                                     (PolyFastA.py:258)     */
  uint32_t sfs_bad;      /* variable columns with < 2 A/C/G/T alleles: getsfs raises IndexError there */
  uint32_t seg_S;        /* synonymous segregating sites                                            */
  uint32_t seg_N;        /* non-synonymous segregating sites                                        */
  uint32_t nstops;       /* codon columns containing a stop codon            (PolyFastA.py:293)     */
  uint32_t missing_cols; /* codon columns without any clean codon; missingcodpos = 3 x this (:305)  */
  uint32_t status;       /* 0                                                                       */
  uint32_t reserved;
} pfa_result;

/* syn_num layout: PFA_SYN_BINS uint32 per unit; bin k (1..64) holds the sum,
 * over codon columns with exactly k distinct clean codons, of the number of
 * synonymous single-nucleotide neighbours of those codons.  The synonymous
 * site count of PolyFastA.py:307 is sum_k syn_num[k] / (3 k).  Bin 0 unused. */
#define PFA_SYN_BINS 65u

typedef struct pfa_plan pfa_plan;

/* bitmap geometry: sites per tile (PFA_TILE_SITES, whatever the arguments) and
 * 64-bit words per tile of every bitmap (4).  tile_off of unit i = sum over
 * earlier units of ceil(n_sites / PFA_TILE_SITES). */
uint32_t pfa_tile_sites(uint32_t max_rows, uint32_t cds);
uint32_t pfa_var_words_per_tile(uint32_t max_rows, uint32_t cds);

/* ---- context -------------------------------------------------------------- */
int pfa_abi_version(void);
/* 16 hex digits: SHA-256 of the sources (csrc/ and this header) the library was compiled from.  The
 * host side compares it with the tree it runs from (polyfasta_amd/_build.py: source_hash) and rebuilds
 * or refuses, so new sources never run against an old library. */
const char* pfa_build_id(void);
int pfa_device_count(void);
int pfa_init(int device);
/* start bringing the HIP runtime up for `device` on a thread of the library's own and return at
 * once (a command line calls this before it loads its own modules; ~0.1 s of hipInit then runs
 * under them); pfa_init() waits for it.  Optional. */
void pfa_prewarm(int device);
/* wait for that thread (a no-op without one): a process that ends before its first pfa_init() must not
 * leave while the thread is inside hipInit */
void pfa_prewarm_join(void);
int pfa_device_index(void); /* the device pfa_init() bound, or -1 */
void pfa_shutdown(void);
const char* pfa_last_error(void);
int pfa_sync(void);

/* ---- device memory (plain device pointers) -------------------------------- */
int pfa_dev_alloc(size_t bytes, void** out);
int pfa_dev_free(void* p);
int pfa_upload(void* dst_dev, const void* src_host, size_t bytes);
/* enqueue the copy on the context stream and return: `src_host` (pinned memory from
 * pfa_host_alloc for a truly asynchronous copy) must stay untouched until pfa_sync() or a
 * pfa_plan_fetch() has returned; scans enqueued afterwards see the uploaded bytes */
int pfa_upload_async(void* dst_dev, const void* src_host, size_t bytes);
int pfa_download(void* dst_host, const void* src_dev, size_t bytes);

/* ---- scan plans ------------------------------------------------------------
 * A plan validates and uploads a descriptor table and owns the device-side
 * output buffers; it can be run any number of times on device-resident bases.
 * Its units are sorted into buckets by the kernel that suits them (row count,
 * coding or not, row-index list or not): one launch per bucket, so a few
 * many-row alignments do not slow a batch of small ones down. */
int pfa_plan_create(const pfa_desc* descs, uint32_t n_desc,
                    const uint32_t* row_idx, size_t n_row_idx,
                    uint64_t n_bases_bytes, uint32_t want, pfa_plan** out);
/* launch on the context stream; asynchronous; returns after enqueue.
 * `variant`: 0 = default kernels (LDS-DMA staging, non-temporal loads: the bases
 * are read once), 2 = the same with the default cache policy (A/B measurements;
 * 256-site-tile launches only) */
int pfa_plan_run(pfa_plan* plan, const uint8_t* d_bases, int variant);
/* record HIP events around the scan launches of every later run (off by default:
 * two event records cost more than a small scan) */
int pfa_plan_set_timing(pfa_plan* plan, int on);
/* elapsed milliseconds (HIP events on the launch stream) of the scan launches of
 * the most recent `n` timed runs, oldest first; returns the count written or <0 */
int pfa_plan_kernel_ms(pfa_plan* plan, float* ms_out, int n);
/* synchronise and copy outputs to host; any pointer may be NULL.
 * sizes: results[n_desc], sfs[sfs_words], syn_num[n_desc*PFA_SYN_BINS],
 *        every bitmap: n_tiles * pfa_var_words_per_tile() uint64 (tile-interleaved, below).
 * The first fetch after a run moves records, SFS and synonymous bins to the host
 * and clears them on the device for the next run; later fetches of the same run
 * are served from the host copy. */
int pfa_plan_fetch(pfa_plan* plan, pfa_result* results, uint32_t* sfs, uint32_t* syn_num,
                   uint64_t* var_bitmap, uint64_t* syn_bitmap, uint64_t* non_bitmap);
/* pfa_plan_run() + pfa_plan_fetch() in one call (one scan per call: the latency path) */
int pfa_plan_run_fetch(pfa_plan* plan, const uint8_t* d_bases, int variant, pfa_result* results, uint32_t* sfs,
                       uint32_t* syn_num, uint64_t* var_bitmap, uint64_t* syn_bitmap, uint64_t* non_bitmap);
/* device pointer of the result records (n_desc x pfa_result) of the last run,
 * valid until that run is fetched (pfa_comm_all_gather_results() reads them) */
int pfa_plan_results_dev(pfa_plan* plan, void** d_results, size_t* bytes);
/* The set of distinct clean codons of every codon column of unit `unit`, in column order — `cods` of
 * getvarCDSsites after its regex filter (PolyFastA.py:291, :301): sets_out[c] has bit (16 e0 + 4 e1 + e2)
 * set (bases A=0 C=1 T=2 G=3, as pfa_classify_codon_mask) for every codon AAA..GGG — stop codons
 * included, :295 removes nothing — that some row of the unit holds at sites 3c..3c+2; 0 = no clean codon
 * (the column is "missing", :305; so is the partial last column of a length that is not a multiple of 3).
 * ceil(n_sites / 3) words.  A re-scan of ONE unit with a kernel of its own, for the rare unit whose
 * synonymous-site count lands on a two-decimal rounding tie: the reference accumulates that count
 * column by column in floating point (:307) and prints round(x, 2) (:178), so on a tie the last bits of
 * its accumulation decide the printed digit, and the host replays the accumulation from these sets
 * (polyfasta_amd/stats.py: syn_sites_replay).  d_bases: the device matrices the plan was run on
 * (NULL = those of its last pfa_plan_run, which must still be resident). */
int pfa_plan_codon_sets(pfa_plan* plan, const uint8_t* d_bases, uint32_t unit, uint64_t* sets_out, uint64_t cap);
uint32_t pfa_plan_n_tiles(const pfa_plan* plan);
/* launch geometry of the plan's largest bucket (by bases read), for reports */
uint32_t pfa_plan_tiles_per_chunk(const pfa_plan* plan);
uint32_t pfa_plan_lds_bytes(const pfa_plan* plan);
uint32_t pfa_plan_rows_per_chunk(const pfa_plan* plan); /* 0: all rows of a tile at once */
uint32_t pfa_plan_tile_width(const pfa_plan* plan);     /* sites per tile: 256, 128 or 64 */
uint32_t pfa_plan_n_launches(const pfa_plan* plan);     /* buckets = kernel launches per run */
/* one line per bucket, largest first: kernel instantiation, units, LDS bytes, chunking
 * (owned by the plan) */
const char* pfa_plan_describe(const pfa_plan* plan);
uint64_t pfa_plan_sfs_words(const pfa_plan* plan);
void pfa_plan_destroy(pfa_plan* plan);

/* one-shot: host bases in, host summaries out (H2D + scan + D2H) */
int pfa_scan_batch(const uint8_t* bases, uint64_t n_bases_bytes,
                   const pfa_desc* descs, uint32_t n_desc,
                   const uint32_t* row_idx, size_t n_row_idx,
                   pfa_result* results, uint32_t* sfs, uint32_t* syn_num,
                   uint64_t* var_bitmap, uint64_t* syn_bitmap, uint64_t* non_bitmap);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------
 * The reference's --dir loop over files (PolyFastA.py:104) is embarrassingly
 * parallel: every rank scans its own share of the alignments, nothing is
 * exchanged on the data path, and the fixed-size integer records are gathered
 * once at the end.  RCCL is loaded on the first of these calls only.  Rank 0
 * makes the unique id; the host language carries its PFA_COMM_ID_BYTES bytes to
 * the other ranks (file, socket, launcher store), then every rank — after
 * pfa_init(device) — calls pfa_comm_init().  Collectives run on the context
 * stream, i.e. behind the scans enqueued before them. */
#define PFA_COMM_ID_BYTES 128
int pfa_comm_unique_id(uint8_t* id_out /* PFA_COMM_ID_BYTES */);
int pfa_comm_init(int rank, int world, const uint8_t* id);
int pfa_comm_rank(void);
int pfa_comm_world(void);
int pfa_comm_barrier(void);                 /* returns when every rank has arrived */
int pfa_comm_all_reduce_max(double* x);     /* in place; e.g. the slowest rank's wall time */
/* `bytes_per_rank` bytes from every rank, in rank order, to every rank (host buffers,
 * staged through device memory; recv_host holds world x bytes_per_rank) */
int pfa_comm_all_gather(const void* send_host, void* recv_host, size_t bytes_per_rank);
/* variable-size gather to ONE rank (the CLI's final gather: only rank 0 prints): rank r sends
 * counts[r] bytes (every rank passes the same `counts`, e.g. from a one-word
 * pfa_comm_all_gather); the root receives them back to back in rank order in recv_host
 * (sum of counts; ignored elsewhere).  ncclSend / ncclRecv: nothing travels to a rank that
 * does not need it, nothing is padded. */
int pfa_comm_gatherv(const void* send_host, uint64_t send_bytes, void* recv_host, const uint64_t* counts, int root);
/* the result records of the plan's last run, of every rank, straight from device
 * memory (no host round trip before the collective); every rank's plan must hold
 * the same number of units; out: world x n_desc records; call it before
 * pfa_plan_fetch() */
int pfa_comm_all_gather_results(pfa_plan* plan, pfa_result* out);
int pfa_comm_destroy(void);

/* Bitmap layout ("tile-interleaved", the order the wavefront ballots produce),
 * the same for the variable-site, synonymous and non-synonymous bitmaps:
 * W = pfa_var_words_per_tile() words per tile (4); bit l
 * of word b of tile t <-> site t*TILE + W*l + b.
 * pfa_bitmap_positions() expands one unit's words to ascending positions;
 * returns the count, or <0 if `cap` is too small. */
int64_t pfa_bitmap_positions(const uint64_t* words, uint32_t n_tiles, uint32_t words_per_tile,
                             uint32_t tile_sites, uint32_t n_sites, uint32_t* pos_out, uint64_t cap);

/* ---- codon-set classifier on the host (same code as the device path) --------
 * `mask`: bit (16*e0 + 4*e1 + e2) set for each distinct codon, bases encoded
 * A=0, C=1, T=2, G=3 (= (ascii >> 1) & 3).  Returns bits 0-2 = codon positions
 * called synonymous, bits 3-5 = non-synonymous (PolyFastA.py:319-434). */
uint32_t pfa_classify_codon_mask(uint64_t mask);
/* the same classification by the wave-uniform form the scan kernels use, on the device, for
 * `n` sets at once (host pointers; used to check the device classifier exhaustively) */
int pfa_classify_codon_masks_dev(const uint64_t* masks, uint32_t n, uint32_t* labels_out);
/* synonymous single-nucleotide neighbours (0..4) of one codon = 3 x syncodfreq
 * (PolyFastA.py:536-557) */
uint32_t pfa_codon_syn_thirds(uint32_t codon);

/* ---- synthetic alignments (bench / tests; polyfasta_amd/synth.py is the spec) */
typedef struct pfa_synth_params {
  uint64_t seed;
  uint64_t first_aln;     /* alignment index of matrix 0                          */
  uint32_t n_aln;         /* number of consecutive matrices                       */
  uint32_t n_rows;
  uint32_t n_sites;
  uint32_t row_stride;    /* bytes; matrix m starts at m * n_rows * row_stride    */
  uint32_t seg_thresh;    /* of 2^24                                              */
  uint32_t tri_thresh;    /* of 2^10                                              */
  uint32_t n_units;
  uint32_t coding;        /* 1: ancestral codons uniform over the 61 sense codons */
} pfa_synth_params;
int pfa_synth_fill_dev(uint8_t* d_bases, const pfa_synth_params* p,
                       const uint32_t* cdf, const uint32_t* units);
int pfa_synth_fill_host(uint8_t* bases, const pfa_synth_params* p,
                        const uint32_t* cdf, const uint32_t* units);

/* ---- FASTA packer (host, readfasta semantics PolyFastA.py:227-250) ----------
 * Parses FASTA text into a row-major upper-case byte matrix.
 * Pass 1: pfa_fasta_index() reports the records; pass 2: pfa_fasta_pack()
 * writes them with the given row stride.  A repeated header resets its record
 * but keeps its first position (dict semantics). */
typedef struct pfa_fasta_rec {
  uint64_t name_off; /* into text: header without '>' and trailing whitespace */
  uint32_t name_len;
  uint32_t seq_len;  /* bytes after rstrip() of every sequence line           */
  uint64_t first_line_off; /* offset of the first sequence line of the LAST occurrence */
  uint64_t end_off;        /* offset just past this record's last line               */
} pfa_fasta_rec;
/* returns number of records (>= 0), PFA_ERR_FORMAT if no header line was seen
 * (the reference's "is not FASTA" case, PolyFastA.py:246-248) */
int64_t pfa_fasta_index(const uint8_t* text, uint64_t n_text, pfa_fasta_rec* recs, uint64_t cap);
int pfa_fasta_pack(const uint8_t* text, uint64_t n_text, const pfa_fasta_rec* recs, uint64_t n_recs,
                   uint8_t* bases_out, uint32_t row_stride);


/* The same records in ONE pass, for texts that may be modified: every record's
 * sequence comes back as one contiguous run [seq_off, seq_off + seq_len) of RAW
 * bytes (not upper-cased; bytes >= 0x80 not looked at) — records written on
 * several lines are compacted in place.  This is what the streaming ingest below
 * does to a file in its pinned ring; upper-casing is left to the card.  Returns
 * the record count, PFA_ERR_FORMAT for "not FASTA", PFA_ERR_ARG when cap is too
 * small (the text may have been modified by then). */
typedef struct pfa_fasta_row {
  uint64_t name_off; /* into text: header without '>' and trailing whitespace */
  uint32_t name_len;
  uint32_t seq_len;
  uint64_t seq_off;  /* into text (after the call) */
} pfa_fasta_row;
int64_t pfa_fasta_index_compact(uint8_t* text, uint64_t n_text, pfa_fasta_row* rows, uint64_t cap);
/* The transport form of a row of the streaming ingest: `n` bases -> ceil(n/4) bytes, two bits per
 * base (A 0, C 1, T 2, G 3 = (byte >> 1) & 3 after case folding; base i of a group of four in bits
 * 2i+1..2i).  Returns 1, or 0 when some byte is not A/C/G/T/a/c/g/t (such a row travels as text:
 * every distinct byte is an allele, PolyFastA.py:258).  The card expands it (repack_kernel). */
int pfa_fasta_pack2(const uint8_t* src, uint64_t n, uint8_t* dst);

/* Batch loader: read, index and pack `n_files` FASTA files with `n_threads`
 * host threads (0 = 16) into `bases_out` (capacity `cap` bytes); matrices are
 * 256-byte aligned, placed in arbitrary order (info[i].base_off), row stride =
 * round_up(n_sites, stride_align).  info[i].status: 0 = packed,
 * PFA_FILE_NOT_FASTA, PFA_FILE_UNEQUAL (sequences of different lengths; the
 * reference prints a comment line for both, PolyFastA.py:135-140,246-248),
 * PFA_FILE_DEFERRED = did not fit this call's buffer (submit it again; when it
 * is larger than `cap`, row_stride holds the needed bytes / 256),
 * PFA_FILE_UNSUPPORTED = non-ASCII bytes in sequence lines (this file only; the
 * others of the call are packed), PFA_ERR_IO = unreadable or not a regular file
 * (e.g. a sub-directory of --dir).  names_out (optional): the record names of file i, each
 * NUL-terminated, back to back from info[i].names_off (--pops matching). */
#define PFA_FILE_NOT_FASTA 1
#define PFA_FILE_UNEQUAL 2
#define PFA_FILE_DEFERRED 3
#define PFA_FILE_UNSUPPORTED 4 /* bytes >= 0x80 in sequence lines: outside one byte per base */
typedef struct pfa_file_info {
  int32_t status;
  uint32_t n_rows;
  uint32_t n_sites;
  uint32_t row_stride;
  uint64_t base_off;  /* into bases_out */
  uint64_t names_off; /* into names_out */
} pfa_file_info;
int pfa_fasta_load_files(const char* const* paths, uint32_t n_files, uint32_t stride_align,
                         uint32_t n_threads, uint8_t* bases_out, uint64_t cap, pfa_file_info* info,
                         char* names_out, uint64_t names_cap, uint64_t* bytes_used,
                         uint64_t* names_used);

/* ---- CSV rows (host; PolyFastA.py:182-198, 502-520) --------------------------
 * Formats the non-coding row of `n` scan units exactly as the reference prints it:
 * "<prefix>S,pi,theta,D\n" with Python's repr() of the doubles, or "<prefix>0,0,0,NA\n"
 * for a unit without variable sites.  prefix i = prefix_blob[prefix_off[i] ..
 * prefix_off[i+1]) is the caller's "file,seqlen,pop,N," text.  a1 / e1 / e2 are the
 * sample-size constants of PolyFastA.py:494-497, 522-533 for row i's N (computed by the
 * caller, so their rounding is the host language's).  Returns the bytes written, or
 * PFA_ERR_ARG; when a row would make the reference raise (math domain error in Dvar)
 * *bad_row is set to its index and nothing is to be used from `out`. */
int64_t pfa_format_plain_rows(const char* prefix_blob, const uint64_t* prefix_off, uint32_t n,
                              const uint32_t* seg_sites, const uint64_t* h_sum, /* pfa_result.seg_sites, .H */
                              const uint32_t* n_rows, const double* seqlen,
                              const double* a1, const double* e1, const double* e2, int jc,
                              char* out, uint64_t cap, int64_t* bad_row);
/* The coding row (PolyFastA.py:151-180): "<name><sites_S>,<sites_N><mid>S_S,S_N,pi_S,pi_N,
 * theta_S,theta_N,D_S,D_N,nstops\n", where name i = "file," and mid i = ",pop,N," are the caller's
 * text, sites_S = sum_k syn_num[k] / (3 k) and sites_N = (seqlen - 3 missing_cols) - sites_S are
 * printed through Python's round(x, 2), and the two polymorphism() tuples use them as their
 * sequence lengths.  syn_num: PFA_SYN_BINS words per row.  sites_s (may be NULL): per row, NaN or the
 * synonymous-site count to use instead of the sum over the bins — the reference's own column-order
 * accumulation replayed for a row on a rounding tie (pfa_plan_codon_sets).  Same return convention as above. */
int64_t pfa_format_cds_rows(const char* name_blob, const uint64_t* name_off, const char* mid_blob, const uint64_t* mid_off,
                            uint32_t n, const pfa_result* res, const uint32_t* syn_num, const double* sites_s, const uint32_t* n_rows,
                            const uint32_t* seqlen, const double* a1, const double* e1, const double* e2, int jc,
                            char* out, uint64_t cap, int64_t* bad_row);
/* repr(float) of CPython 3 into out32 (at least 32 bytes, not NUL-terminated);
 * returns the length */
int pfa_format_double(double x, char* out32);

/* ---- streaming ingest: files -> device-resident matrices -> launched plans -----
 * The feeder of the scan for the reference's file loop (PolyFastA.py:104-140,
 * readfasta :227-250, population split :122-134), built so that the PCIe link is
 * the only thing a directory of alignments waits for:
 *   loader threads   read() each file ONCE, index it in place
 *                    (pfa_fasta_index_compact), match the --pops keys against its
 *                    record names, and copy its rows in transport form — two bits
 *                    per base for clean A/C/G/T rows (pfa_fasta_pack2), the bytes
 *                    otherwise — into a ring of pinned slots, in feed order;
 *   uploader thread  copies every slot to the card as soon as its files have
 *                    landed (own HIP stream, at most two copies in flight);
 *   the caller       (pfa_ingest_next) takes whatever has been uploaded, lets the
 *                    card expand the rows into row-major upper-case matrices
 *                    (repack_kernel: one pass at HBM speed), builds the scan plan
 *                    of those files and launches it.
 * Files are placed in the ring as their loaders finish, so a batch is a SET of feed
 * positions (ascending inside the batch), not a run of them: a caller that prints in
 * feed order holds rows back until the positions before them have arrived.
 * Nothing here scans on the CPU. */
typedef struct pfa_ingest pfa_ingest;
typedef struct pfa_ingest_opts {
  uint32_t n_threads;     /* loader threads; 0 = min(16, CPUs of this process / local ranks) */
  uint32_t stride_align;  /* row stride = round_up(n_sites, stride_align); 0 = 128 */
  uint32_t unit_flags;    /* PFA_FLAG_CDS or 0: flags of every scan unit */
  uint32_t n_keys;        /* --pops keys (0: one unit per file) */
  const char* const* keys;/* NUL-terminated byte strings, matched as substrings of the record names */
  uint64_t slot_bytes;    /* size of a pinned slot = one H2D copy; 0 = 16 MiB (less for small inputs) */
  uint32_t n_slots;       /* pinned slots; 0 = 8 */
  uint32_t reserved;
} pfa_ingest_opts;

#define PFA_NO_KEY 0xFFFFFFFFu
#define PFA_BATCH_REGULAR 1u /* every file packed, every key found, coding lengths multiples of 3 */

/* One batch = some files of the feed order, ascending.  Every pointer is owned
 * by the ingest and valid until the next pfa_ingest_next() / pfa_ingest_close(). */
typedef struct pfa_ingest_batch {
  uint32_t n_files;
  uint32_t n_units;          /* scan units = CSV rows */
  uint32_t first_pos;        /* lowest feed position in the batch */
  uint32_t flags;            /* PFA_BATCH_* */
  const uint32_t* file_idx;  /* [n_files] index into the `paths` of pfa_ingest_open() */
  const int32_t* status;     /* [n_files] 0 = packed, PFA_FILE_NOT_FASTA, PFA_FILE_UNEQUAL, PFA_ERR_IO */
  const uint32_t* n_rows;    /* [n_files] records */
  const uint32_t* n_sites;   /* [n_files] alignment length (status 0) */
  const uint32_t* key_rows;  /* [n_files x n_keys] records whose name holds key k (0: "not found") */
  const uint32_t* unit_file; /* [n_units] file of the unit, as an index into THIS batch's arrays */
  const uint32_t* unit_key;  /* [n_units] key index, PFA_NO_KEY without --pops */
  const uint32_t* unit_rows; /* [n_units] N of the unit */
  const uint32_t* bad_bytes; /* [n_files] nonzero once the plan has been fetched: bytes >= 0x80 in
                                sequence lines (that file is PFA_FILE_UNSUPPORTED; a file whose records differ
                                in BYTE length and hold such bytes has that status already: the reference counts
                                characters, PolyFastA.py:239-245, and the caller reads such files as text) */
  pfa_plan* plan;            /* already launched; NULL when n_units == 0; owned by the ingest */
} pfa_ingest_batch;

/* `paths`: every file the job may read (kept: names for the rows).  Nothing is read
 * before pfa_ingest_feed(). */
int pfa_ingest_open(const char* const* paths, uint32_t n_files, const pfa_ingest_opts* opts, pfa_ingest** out);
/* The reference's --dir listing done here (PolyFastA.py:93-94, :104): every entry of `dir` except "." and ".." as
 * "dir/name", sorted as Python sorts those strings (for UTF-8 names byte order is code-point order).
 * PFA_ERR_UNSUPPORTED when a name is not UTF-8 (the caller lists and sorts itself), PFA_ERR_IO when the
 * directory cannot be opened. */
int pfa_ingest_open_dir(const char* dir, const pfa_ingest_opts* opts, pfa_ingest** out, uint32_t* n_files);
/* The same job from a listing somebody else made (PolyFastA.py:93-94, :104): file f is "dir/" + names_blob[names_off[f] ..
 * names_off[f+1]), in the order given (pfa_ingest_names() of the ingest that listed `dir`).  The ranks of a multi-GPU job
 * take rank 0's listing this way: R processes walking one directory of 100 k entries at once serialise on its lock
 * (26 ms for 4 ranks, 51 ms for 6 on a 16 384-file tmpfs directory; 5 ms alone). */
int pfa_ingest_open_names(const char* dir, const char* names_blob, const uint64_t* names_off, uint32_t n_files,
                          const pfa_ingest_opts* opts, pfa_ingest** out);
/* path i of the ingest (owned by it) */
const char* pfa_ingest_path(const pfa_ingest* ing, uint32_t i);
/* append files (indices into `paths`; NULL = 0..n-1) to the feed order; `last` != 0: no
 * further feed follows (a multi-GPU rank feeds what it claims from the shared queue) */
int pfa_ingest_feed(pfa_ingest* ing, const uint32_t* file_idx, uint32_t n, int last);
/* files fed but not yet handed out in a batch */
uint32_t pfa_ingest_backlog(const pfa_ingest* ing);
/* blocks until at least one uploaded slot is there; 1 = batch, 0 = every fed file has been
 * handed out and the last feed was marked `last`, < 0 = error */
int pfa_ingest_next(pfa_ingest* ing, pfa_ingest_batch* out);
/* [0] seconds the loader threads spent in read(), [1] indexing, packing rows and copying them into
 * the ring, [2] bytes read, [3] seconds the uploader had copies in flight, [4] bytes copied
 * (transport bytes), [5] loader threads, [6] seconds the loader threads waited for their turn or a
 * free slot, [7] batches */
int pfa_ingest_stats(const pfa_ingest* ing, double* out8);
/* display names (the text after the last '/', PolyFastA.py:110) of every path of
 * pfa_ingest_open(): name f = blob[off[f] .. off[f+1]); owned by the ingest */
int pfa_ingest_names(const pfa_ingest* ing, const char** blob, const uint64_t** off);
void pfa_ingest_close(pfa_ingest* ing);

/* CSV rows of units named by index: row i is file `file_idx[i]` (name =
 * names_blob[names_off[f] .. names_off[f+1])), population key `key[i]` (PFA_NO_KEY:
 * "NA").  coding == 0: the 8-column row (pfa_format_plain_rows), else the 14-column
 * one (pfa_format_cds_rows; syn_num: PFA_SYN_BINS words per row; sites_s: its optional per-row
 * override, may be NULL).  Same return convention as those two. */
int64_t pfa_format_rows_indexed(const char* names_blob, const uint64_t* names_off,
                                const char* keys_blob, const uint64_t* keys_off,
                                uint32_t n, const uint32_t* file_idx, const uint32_t* key,
                                const uint32_t* n_rows, const uint32_t* n_sites,
                                const pfa_result* res, const uint32_t* syn_num, const double* sites_s,
                                const double* a1, const double* e1, const double* e2, int jc, int coding,
                                char* out, uint64_t cap, int64_t* bad_row);

/* pinned (page-locked) host memory for the ingest buffer: pre-faulted, and H2D
 * from it runs at the PCIe rate */
int pfa_host_alloc(size_t bytes, void** out);
int pfa_host_free(void* p);

/* CPUs this process may keep busy — what the ingest sizes its loader pool by and what decides whether host waits
 * spin or sleep: the CPUs it may run on and the CPU-time quota of its control group (cgroup v2 cpu.max / v1
 * cpu.cfs_quota_us, the tightest from its group up to the root), shared out over the LOCAL_WORLD_SIZE ranks of the
 * node; PFA_CPUS overrides.  No reference counterpart (PolyFastA.py:104 is one thread); it is here because one rank
 * per GPU inside a container's CPU quota is how `--dir` runs on a shared node.
 * With both arguments NULL: that number.  With a cgroup mount point and a /proc/<pid>/cgroup-style file: the quota
 * found there alone, in CPUs (1e9 = none) — for tests. */
double pfa_host_cpu_share(const char* cgroup_root, const char* proc_cgroup);
/* The loader threads of a single process's ingest are spread round robin over the last-level-cache domains of the CPUs the
 * process may run on (a CCD of an EPYC: 8 cores and their sibling threads), each free to move inside its domain: left to the
 * scheduler, 16 loaders now and then crowd onto a few domains or onto sibling threads, and the files -> CSV rate of a run
 * swings by 10-40 % (profiles/r04_loader_l3_spread_ab.txt).  Not for the ranks of a multi-GPU job (measured slightly slower
 * there) nor for a mask that spans sockets.  PFA_PIN_LOADERS=0 switches it off, =l3 forces it.  No reference
 * counterpart (one thread).  This entry reports the grouping — for tests: `cpu_sys_root` = a /sys/devices/system/cpu-style
 * tree (cpu<N>/cache/index3/id), NULL = the machine's; domain_of[i] = domain index (in mask order) of the i-th CPU of
 * the affinity mask, for at most `cap` CPUs.  Returns the number of domains, 0 when sysfs does not say. */
int pfa_host_l3_domains(const char* cpu_sys_root, int* domain_of, int cap);

#ifdef __cplusplus
}
#endif
#endif /* PFA_H */
