// CPU-side pieces under -fsanitize=address,undefined: the KMC listing reader of the product (kmcex_amd/csrc/kmc_reader.cpp:
// open, count_classes, next_batch with several threads, raw record copies) and the oracle (build from that listing, save,
// load, query).  Prints a few figures the test compares with the goldens; any sanitizer report makes the run fail.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../kmcex_amd/csrc/kmc_reader.h"
extern "C" {
#include "../oracle/kmx_oracle.h"
}

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	const std::string db = argv[1], out_dir = argv[2], queries = argv[3];
	kmx::KmcListing l;
	if (!l.open(db)) { printf("open failed: %s\n", l.error().c_str()); return 3; }
	l.set_threads(3);
	const int W = l.words();
	uint64_t nbf[3], bad = 0, skipped = 0;
	l.count_classes(1, 1023, 1, nbf, &bad, &skipped);
	std::vector<uint64_t> km(l.kmer_count() * W + 1);
	std::vector<uint32_t> cnt(l.kmer_count() + 1);
	size_t n = 0;
	for (size_t got; (got = l.next_batch(km.data() + n * W, cnt.data() + n, 7001)) > 0;) n += got;   // ragged batches
	std::vector<unsigned char> raw(l.records() * l.record_bytes());
	l.copy_records(0, l.records(), raw.data());
	l.copy_records(5, 3, raw.data());                                 // an inner range
	kmx::KmcListing missing;
	if (missing.open(db + "_nope")) return 4;
	printf("listed %zu of %llu, n_bf0 %llu, bad %llu, skipped %llu\n", n, (unsigned long long)l.kmer_count(), (unsigned long long)nbf[0], (unsigned long long)bad, (unsigned long long)skipped);
	kmo_model *m = kmo_create(1, 1023, 7, 5);
	if (!m || kmo_build(m, (int)l.kmer_length(), km.data(), cnt.data(), n, l.kmer_count())) return 5;
	if (kmo_save(m, out_dir.c_str())) return 6;
	kmo_model *m2 = kmo_load(out_dir.c_str());
	if (!m2) return 7;
	std::vector<char> q;
	std::vector<std::string> lines;
	FILE *f = fopen(queries.c_str(), "r");
	if (!f) return 8;
	char buf[256];
	while (fgets(buf, sizeof buf, f) && lines.size() < 2000) { buf[strcspn(buf, "\r\n")] = 0; if (buf[0]) lines.push_back(buf); }
	fclose(f);
	const int len = (int)lines[0].size();
	for (auto &s : lines) q.insert(q.end(), s.begin(), s.begin() + len);
	std::vector<int32_t> occ(lines.size());
	if (kmo_query_ascii(m2, q.data(), len, len, lines.size(), occ.data(), 2)) return 9;
	long long sum = 0;
	for (int v : occ) sum += v;
	kmo_stats st;
	kmo_get_stats(m, &st);
	printf("attempts %llu successes %llu rest %llu occ_sum_first_%zu %lld\n", (unsigned long long)st.attempts, (unsigned long long)st.successes, (unsigned long long)st.rest_entries, lines.size(), sum);
	kmo_destroy(m);
	kmo_destroy(m2);
	return 0;
}
