// include/kmodel.hpp -- drop-in C++ facade with the reference's own names over the C ABI (include/kmx.h).
//
// A program written against lzhLab/kmcEx's kmodel.hpp (README.md:64-93, main.cpp:143-149) compiles against
// this header and links libkmx.so instead of pulling in the header-only CPU implementation (tests/facade_query.cpp is
// written the way those snippets are: unqualified std names, Tools::get_file_name, KModel() + load(dir)):
//
//     KModel* km = get_model(ci, cs, n_hash, n_bit);      // kmodel.hpp:674
//     km->init(kmc_database);                              // kmodel.hpp:57   (README: init_KModel)
//     km->save(dir);                                       // kmodel.hpp:173  (README: save_model)
//     KModel* km2 = get_model(dir);                        // kmodel.hpp:680
//     std::vector<int> occ = km2->kmer_to_occ(kmers, 4);   // kmodel.hpp:90
//
// Error behaviour follows the reference: a message on stdout and exit(1) (kmodel.hpp:394-397, :682-685).
// Unlike the reference the object has a destructor, so the device memory can be released.
#pragma once
#ifndef KMODEL_H
#define KMODEL_H

// Everything the reference header makes visible to its includers (kmodel.hpp:6-21): main.cpp uses ifstream, strncmp,
// sprintf and system without including their headers itself.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "kmx.h"

// kmc_api/kmer_defs.h:27-28 (pulled in through kmc_api/kmc_file.h, kmodel.hpp:14): main.cpp's entry point is `_tmain`
#ifndef _WIN32
#ifndef _tmain
#define _TCHAR char
#define _tmain main
#endif
#endif

// The reference header leaks `using namespace std;` and its README snippets and main.cpp rely on it (unqualified string,
// vector<string>, cout); a drop-in has to leak it too.
using namespace std;

const string BASE_CHAR = "ACGT";          // kmodel.hpp:23-24 (visible to includers of the reference header)
const int BLOACK_SIZE = 1 << 19;

// the one helper of tools.hpp that main.cpp uses outside the class (tools.hpp:102-105)
class Tools {
public:
	static string get_file_name(string path)
	{
		const size_t pos = path.find_last_of('/');
		return pos == string::npos ? path : path.substr(pos + 1);
	}
};

class KModel {
public:
	KModel() : h_(nullptr) { abi(); }                                  // kmodel.hpp:43; fill it with load(dir)
	explicit KModel(kmx_model *h) : h_(h) { abi(); }
	~KModel() { kmx_destroy(h_); }
	// a binary built against an older kmx.h must not hand its structs to a newer libkmx.so
	static void abi()
	{
		if (kmx_abi_version() != KMX_ABI_VERSION) {
			std::cout << "libkmx.so has ABI version " << kmx_abi_version() << ", this program was built for " << KMX_ABI_VERSION << "; rebuild it" << std::endl;
			exit(1);
		}
	}
	KModel(const KModel &) = delete;
	KModel &operator=(const KModel &) = delete;

	// kmodel.hpp:57 -- two passes over the KMC listing, rest-table build
	// kmodel.hpp:57-86.  KMX_DEVICES=0,1,2,... (HIP device numbers, repeats allowed): the model is built by those GPUs
	// together, one host thread per device inside libkmx.so, and this object keeps the replica of the first one.
	// KMX_PARTITION=range: every coupled array cut by position range over the devices, the words of a round written
	// straight into the owners' memory through peer mappings (kmx_build_from_kmc_multi_ex, KMX_PARTITION_RANGE);
	// KMX_PARTITION=ring (the default): arrays owned whole, lists travel with peer copies.  Unset: the current device alone.
	// A value that does not parse is an error, like a bad argument of the reference's own command line.
	void init(std::string db_file)
	{
		std::vector<int> devs;
		if (const char *e = std::getenv("KMX_DEVICES")) {
			for (const char *p = e; *p;) {
				char *end = nullptr;
				const long v = std::strtol(p, &end, 10);
				if (end == p || v < 0 || (*end != ',' && *end != 0) || (*end == ',' && end[1] == 0)) {
					std::cout << "KMX_DEVICES=" << e << ": a comma-separated list of HIP device numbers is expected" << std::endl;
					exit(1);
				}
				devs.push_back((int)v);
				p = *end == ',' ? end + 1 : end;
			}
		}
		int partition = KMX_PARTITION_RING;
		if (const char *e = std::getenv("KMX_PARTITION")) {
			const std::string v(e);
			if (v == "range") partition = KMX_PARTITION_RANGE;
			else if (v == "range-rccl") partition = KMX_PARTITION_RANGE_RCCL;     // the same partition, the words as fixed-size RCCL messages (one device per entry of KMX_DEVICES)
			else if (v != "ring" && !v.empty()) { std::cout << "KMX_PARTITION=" << v << ": ring, range or range-rccl" << std::endl; exit(1); }
		}
		if (devs.empty()) { check(kmx_build_from_kmc(h_, db_file.c_str())); return; }
		kmx_stats st;
		check(kmx_get_stats(h_, &st));
		std::vector<kmx_model *> hs(devs.size(), nullptr);
		for (size_t d = 0; d < devs.size(); d++) {
			const int rc = kmx_create_on(devs[d], st.ci, st.cs, st.nh, st.nb, &hs[d]);
			if (rc) { for (size_t e2 = 0; e2 < d; e2++) kmx_destroy(hs[e2]); check(rc); }
		}
		const int rc = kmx_build_from_kmc_multi_ex(hs.data(), (int)hs.size(), db_file.c_str(), partition);
		if (rc) { for (kmx_model *h : hs) kmx_destroy(h); check(rc); }
		kmx_destroy(h_);
		h_ = hs[0];
		for (size_t d = 1; d < hs.size(); d++) kmx_destroy(hs[d]);
	}
	void init_KModel(std::string db_file) { init(db_file); }                 // README.md:76

	// kmodel.hpp:90 -- t_num is accepted for source compatibility; the batch runs on the GPU
	std::vector<int> kmer_to_occ(std::vector<std::string> kmer_v, int t_num = 4)
	{
		(void)t_num;
		const size_t n = kmer_v.size();
		std::vector<int> occ_v(n);
		if (!n) return occ_v;
		static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bits on every supported target");
		// the reference answers every string on its own, whatever its length.  One length (the usual batch): the strings
		// go to the library as they lie, and it cuts them into chunks that worker threads pack while the GPU answers the
		// chunk before (kmx_query_strings).  Mixed lengths: one such call per length.
		const size_t len0 = kmer_v[0].size();
		bool uniform = true;
		for (size_t i = 1; i < n && uniform; i++) uniform = kmer_v[i].size() == len0;
		std::vector<const char *> ptrs;
		if (uniform) {
			ptrs.resize(n);
			for (size_t i = 0; i < n; i++) ptrs[i] = kmer_v[i].data();
			check(kmx_query_strings(h_, ptrs.data(), (int)len0, n, (int32_t *)occ_v.data()));
			return occ_v;
		}
		std::vector<size_t> order(n);
		for (size_t i = 0; i < n; i++) order[i] = i;
		std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return kmer_v[a].size() < kmer_v[b].size(); });
		for (size_t lo = 0; lo < n;) {
			const size_t len = kmer_v[order[lo]].size();
			size_t hi = lo;
			ptrs.clear();
			while (hi < n && kmer_v[order[hi]].size() == len) ptrs.push_back(kmer_v[order[hi++]].data());
			std::vector<int32_t> part(hi - lo);
			check(kmx_query_strings(h_, ptrs.data(), (int)len, hi - lo, part.data()));
			for (size_t j = lo; j < hi; j++) occ_v[order[j]] = part[j - lo];
			lo = hi;
		}
		return occ_v;
	}

	// kmodel.hpp:100
	int kmer_to_occ(std::string kmer, uint32_t r_occ = 0)
	{
		(void)r_occ;
		int32_t occ = 0;
		check(kmx_query_ascii(h_, kmer.data(), (int)kmer.size(), (int)kmer.size(), 1, &occ));
		return occ;
	}

	void save(std::string save_dir) { check(kmx_save(h_, save_dir.c_str())); }       // kmodel.hpp:173
	void save_model(std::string save_dir) { save(save_dir); }                          // README.md:78

	// kmodel.hpp:209 -- the model directory replaces whatever this object held (parameters come from its header)
	void load(std::string save_dir)
	{
		kmx_model *h = nullptr;
		check(kmx_load(save_dir.c_str(), &h));
		kmx_destroy(h_);
		h_ = h;
	}
	void load_model(std::string save_dir) { load(save_dir); }

	// kmodel.hpp:118, :127 -- same table, figures from kmx_get_stats
	void show_header_info()
	{
		kmx_stats st;
		kmx_get_stats(h_, &st);
		std::cout << "KMCEX:" << std::endl;
		std::cout << "   kmodel number hash                 :     " << st.nh << std::endl;
		std::cout << "   kmodel bit array                   :     " << st.nb << std::endl;
		std::cout << "   total kmercount                    :     " << st.n_total << std::endl;
		std::cout << "   kmercount in blommfilter           :     " << st.n_total - st.n_km << std::endl;
		std::cout << "   kmercount in kmodel                :     " << st.n_km << std::endl;
	}
	void show_kmodel_info()
	{
		kmx_stats st;
		kmx_get_stats(h_, &st);
		uint64_t bf = 0;
		for (int i = 0; i < st.bf_num; i++) bf += st.byte_bf[i] + st.byte_bf_back[i];
		const uint64_t km = 2 * st.km_byte_size * (uint64_t)st.nb, mb = 1024 * 1024;
		double total_s = 0;
		kmx_last_build_seconds(h_, nullptr, &total_s);
		std::cout << "   kmercount hash map                 :     " << st.rest_entries << std::endl;
		std::cout << "   memory bloomfilter                 :     " << bf / mb << "MB" << std::endl;
		std::cout << "   memory bit array                   :     " << km / mb << "MB" << std::endl;
		std::cout << "   memory rest map                    :     " << st.rest_bytes / mb << "MB" << std::endl;
		std::cout << "   total memory                       :     " << (bf + km + st.rest_bytes + st.byte_km_back) / mb << "MB" << std::endl;
		std::cout << "   build time cost                    :     " << total_s << std::endl;
	}

	kmx_model *handle() { return h_; }

private:
	static void die(const char *msg)
	{
		std::cout << msg << std::endl;
		std::exit(1);
	}
	static void check(int rc)
	{
		if (rc != KMX_OK) die(kmx_last_error());
	}
	kmx_model *h_;
};

// kmodel.hpp:674-677
inline KModel *get_model(int ci = 1, int cs = 1023, int num_hash = 7, int num_bit = 5)
{
	kmx_model *h = nullptr;
	if (kmx_create(ci, cs, num_hash, num_bit, &h) != KMX_OK) {
		std::cout << kmx_last_error() << std::endl;
		std::exit(1);
	}
	return new KModel(h);
}

// kmodel.hpp:680-696 (load_model in the README's vocabulary)
inline KModel *get_model(std::string save_dir)
{
	kmx_model *h = nullptr;
	if (kmx_load(save_dir.c_str(), &h) != KMX_OK) {
		std::cout << kmx_last_error() << std::endl;
		std::exit(1);
	}
	return new KModel(h);
}
inline KModel *load_model(std::string save_dir) { return get_model(save_dir); }

#endif
