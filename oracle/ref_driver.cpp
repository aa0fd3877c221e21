// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin command-line driver around the *real* reference implementation.  It is compiled
// against the reference sources where they lie under /root/reference (see oracle/Makefile);
// no reference source is copied into this repository.  The resulting binary lives in
// oracle/_ref/ (git-ignored) and is used (a) to pin oracle/kmx_oracle.c against the true
// reference and (b) to generate the golden vectors under tests/golden/.
//
// Commands
//   build  <kmc_db_prefix> <out_dir> <ci> <cs> <nh> <nb>     get_model(ci,cs,nh,nb)->init(db)->save(dir)
//                                                            (kmodel.hpp:674, :57, :173)
//   query  <model_dir> <kmers.txt> <out.txt> [t_num]         get_model(dir)->kmer_to_occ(vector<string>,t_num)
//                                                            (kmodel.hpp:680, :90)
//   hash   <string> <seed_index>                             Tools::murmur_hash64 (tools.hpp:16)
//   minkmer <string>                                         Tools::get_min_kmer  (tools.hpp:160)
//   occubin <max_counter> <nh>                               OccuBin table dump   (occu_bin.hpp:27-83)
//   list   <kmc_db_prefix> <out.txt>                         CKMCFile listing     (kmc_file.cpp:428)
#include "kmodel.hpp"
#include <cstdio>
#include <cstdlib>
#include <chrono>

static int usage() {
	fprintf(stderr, "usage: ref_driver build|query|hash|minkmer|occubin|list ...\n");
	return 2;
}

int main(int argc, char **argv) {
	if (argc < 2) return usage();
	std::string cmd = argv[1];
	if (cmd == "build" && argc == 8) {
		KModel *km = get_model(atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
		auto t0 = std::chrono::high_resolution_clock::now();
		km->init(argv[2]);
		std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
		fprintf(stderr, "init_seconds %.6f\n", dt.count());
		km->show_kmodel_info();
		km->save(argv[3]);
		return 0;
	}
	if (cmd == "query" && (argc == 5 || argc == 6)) {
		int t_num = argc == 6 ? atoi(argv[5]) : 4;
		KModel *km = get_model(std::string(argv[2]));
		std::vector<std::string> v;
		{
			std::ifstream fin(argv[3]);
			std::string s;
			while (std::getline(fin, s)) v.push_back(s);
		}
		auto t0 = std::chrono::high_resolution_clock::now();
		std::vector<int> r = km->kmer_to_occ(v, t_num);
		std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
		fprintf(stderr, "query_seconds %.6f n %zu t_num %d\n", dt.count(), v.size(), t_num);
		FILE *fo = fopen(argv[4], "w");
		for (size_t i = 0; i < r.size(); i++) fprintf(fo, "%d\n", r[i]);
		fclose(fo);
		return 0;
	}
	if (cmd == "hash" && argc == 4) {
		std::string s = argv[2];
		int si = atoi(argv[3]);
		printf("%016llx\n", (unsigned long long)Tools::murmur_hash64(s.c_str(), (int)s.size(), HashSeeds[si]));
		return 0;
	}
	if (cmd == "minkmer" && argc == 3) {
		printf("%s\n", Tools::get_min_kmer(argv[2]).c_str());
		return 0;
	}
	if (cmd == "occubin" && argc == 4) {
		int mc = atoi(argv[2]), nh = atoi(argv[3]);
		OccuBin ob(mc, nh);
		for (int occ = 0; occ < mc; occ++) {
			uint32_t b = ob.occ_to_bin((uint32_t)occ);
			printf("%d %u %u\n", occ, b, ob.bin_to_mean(b));
		}
		return 0;
	}
	if (cmd == "list" && argc == 4) {
		CKMCFile db;
		if (!db.OpenForListing(argv[2])) { fprintf(stderr, "cannot open db\n"); return 1; }
		CKmerAPI ko(db.KmerLength());
		uint32 c;
		FILE *fo = fopen(argv[3], "w");
		fprintf(fo, "# k %u total %llu\n", db.KmerLength(), (unsigned long long)db.KmerCount());
		while (db.ReadNextKmer(ko, c)) fprintf(fo, "%s %u\n", ko.to_string().c_str(), c);
		fclose(fo);
		return 0;
	}
	return usage();
}
