/* oracle/kmx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the kmcEx KModel insert/query path, used as the parity checker for
 * the HIP product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (kmcex_amd/csrc, libkmx.so) never links or calls it.
 *
 * Parity status: PINNED.  Every function is checked bit-for-bit against the real reference
 * compiled from /root/reference (oracle/_ref/ref_driver, see oracle/Makefile) by
 * tests/golden/make_golden.py, and against the committed golden vectors in tests/golden/.
 *
 * Packed k-mer layout: W = ceil(k/32) uint64 words per k-mer, word 0 most significant, 2k-bit
 * integer right-aligned, A=0 C=1 G=2 T=3, first base most significant (tools.hpp:63-76).
 */
#ifndef KMX_ORACLE_H
#define KMX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kmo_model kmo_model;

typedef struct kmo_stats {
	uint64_t n_total;        /* k-mers listed                                   */
	uint64_t n_km;           /* k-mers routed to the coupled arrays             */
	uint64_t n_bf[3];        /* k-mers per Bloom filter class                   */
	uint64_t attempts;       /* insert_to_array calls (kmodel.hpp:590)          */
	uint64_t successes;      /* ... that returned true                          */
	uint64_t rest_entries;   /* rows in rest.bin (incl. the Q1 duplicate rows)  */
	uint64_t km_byte_size;   /* bytes per tag / value array                     */
	uint64_t byte_km_back;
	uint64_t byte_bf[3], byte_bf_back[3];
} kmo_stats;

/* primitives (KAT surface) */
uint64_t kmo_murmur64(const void *key, int len, uint32_t seed);           /* tools.hpp:16-50   */
uint32_t kmo_hash_seed(int index);                                         /* tools.hpp:9       */
void     kmo_min_kmer(const char *s, int len, char *out);                  /* tools.hpp:160-167 */
int      kmo_occubin_table(int max_counter, int nh, uint32_t *bin_of_occ, uint32_t *mean_of_bin); /* occu_bin.hpp:27-83 */
void     kmo_packed_to_ascii(const uint64_t *words, int k, char *out);

/* model life cycle (kmodel.hpp:674-696, :57-86, :173-235) */
kmo_model *kmo_create(int ci, int cs, int nh, int nb);
void       kmo_destroy(kmo_model *m);
/* kmers: n*W words in LISTING ORDER, counts already within [min,max]; total_kmers = KmerCount() */
int        kmo_build(kmo_model *m, int k, const uint64_t *kmers, const uint32_t *counts, uint64_t n, uint64_t total_kmers);
int        kmo_build_declared(kmo_model *m, int k, const uint64_t *kmers, const uint32_t *counts, uint64_t n, const uint64_t n_bf[3], uint64_t total_kmers);   /* a prefix of a build on arrays of its full size */
int        kmo_save(const kmo_model *m, const char *dir);
kmo_model *kmo_load(const char *dir);
void       kmo_get_stats(const kmo_model *m, kmo_stats *st);

/* the same build in per-rank pieces (checks the multi-GPU protocol of kmcex_amd/dist.py on the CPU; tests only) */
int      kmo_shard_begin(kmo_model *m, int k, const uint64_t n_bf[3], uint64_t total_kmers);
uint64_t kmo_shard_classify(kmo_model *m, const uint64_t *kmers, const uint32_t *counts, uint64_t n, uint64_t *out_kmers, uint32_t *out_counts);
int      kmo_ring_insert(kmo_model *m, int a, const uint64_t *kmers, const uint32_t *counts, int n, uint64_t *out_kmers, uint32_t *out_counts);
int      kmo_shard_complete(kmo_model *m, const uint64_t *kmers, const int32_t *counts, uint64_t n, uint64_t attempts, uint64_t successes);

/* query (kmodel.hpp:90-116).  strs: n records of `stride` bytes, each holding `len` chars. */
int kmo_query_ascii(const kmo_model *m, const char *strs, int len, int stride, uint64_t n, int32_t *out, int threads);
int kmo_query_packed(const kmo_model *m, int k, const uint64_t *kmers, uint64_t n, int32_t *out, int threads);

/* raw views for byte-level comparison with the product (on-disk layout, kmodel.hpp:183-202) */
const uint8_t *kmo_bf(const kmo_model *m, int i);
const uint8_t *kmo_bf_back(const kmo_model *m, int i);
const uint8_t *kmo_km_back(const kmo_model *m);
const uint8_t *kmo_value_array(const kmo_model *m, int a);   /* bit_array_1 */
const uint8_t *kmo_tag_array(const kmo_model *m, int a);     /* bit_array_2 */

#ifdef __cplusplus
}
#endif
#endif
