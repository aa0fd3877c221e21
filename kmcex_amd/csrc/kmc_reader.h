// kmc_reader.h -- from-scratch listing reader for KMC databases (host C++).
//
// Replaces, for the KModel build path only, what the reference gets from its vendored KMC 3.1.0 API:
//   CKMCFile::OpenForListing / ReadNextKmer / RestartListing / KmerCount / KmerLength
//   (kmc_file.cpp:66-99, :428-515, :647-664, :763, :740) and CKmerAPI::to_string (kmer_api.h:433).
// Written from the on-disk format; it decodes records straight into packed
// 2-bit k-mers (W = ceil(k/32) u64 words, word 0 most significant) instead of strings, whole batches at
// a time, so that the host feed is a memcpy-speed loop rather than the reference's per-k-mer string build.
#pragma once
#include <cstdint>
#include <atomic>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace kmx {

// a vector whose resize() does not zero what is about to be overwritten (the LUTs of a KMC2 database with 512 bins are 67 MB)
template <typename T> struct NoInitAlloc : std::allocator<T> {
	template <typename U> struct rebind { typedef NoInitAlloc<U> other; };
	template <typename U> void construct(U *p) noexcept { ::new ((void *)p) U; }
	template <typename U, typename... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<uint64_t, NoInitAlloc<uint64_t>> LutVec;

class KmcListing {
public:
	~KmcListing() { close(); }
	// false + error() on failure.  Supports KMC1 (version 0) and KMC2 (0x200) prefix files, mode 0 counters.
	// load_lut = false: the LUT(s) stay in the file until lut() or read_lut() asks for them (the GPU decoder wants them in ITS
	// pinned memory: a KMC2 database with 512 bins has 67 MB of them, and a fresh host copy costs its page faults twice)
	bool open(const std::string &prefix, bool load_lut = true);
	void close();
	void restart();                                  // RestartListing
	uint32_t kmer_length() const { return k_; }
	uint64_t kmer_count() const { return total_; }   // KmerCount() with untouched min/max
	uint32_t min_count() const { return min_count_; }
	uint64_t max_count() const { return max_count_; }
	int words() const { return (int)((k_ + 31) / 32); }
	// Next batch in listing order; counts outside [min_count, max_count] are skipped exactly like
	// ReadNextKmer does (kmc_file.cpp:513).  Returns the number of k-mers produced (0 at the end).
	size_t next_batch(uint64_t *kmers, uint32_t *counts, size_t max_n);
	void set_threads(int t) { threads_ = t < 1 ? 1 : t; }       // decode threads per batch (records are fixed-size)
	// Pass 1 of KModel::init (kmodel.hpp:423-428) without materialising k-mers: number of listed k-mers per count
	// ci+i (i < bf_num) and the number of listed counts outside [ci, cs].  Does not move the listing cursor.
	// `not_listed` (optional): records whose count lies outside the header's [min_count, max_count] (ReadNextKmer skips them).
	// `threads` (0: as set_threads): the scan is bound by the copy out of the page cache, more threads than a decode wants pay off.
	void count_classes(uint32_t ci, uint32_t cs, int bf_num, uint64_t n_bf[3], uint64_t *out_of_range, uint64_t *not_listed = nullptr, int threads = 0) const;
	const std::string &error() const { return err_; }
	// true once any read of the record file has failed (EIO, a file truncated after open): whatever the calls that hit it
	// produced must not be used -- next_batch / copy_records / count_classes keep going with stale bytes, the callers check this
	bool io_failed() const { return io_failed_.load(std::memory_order_relaxed); }
	// Raw access for a decoder that runs elsewhere (the GPU: k_kmc_decode): fixed-size records [suffix bytes, big-endian |
	// counter bytes, little-endian], the concatenated LUT(s) with a sentinel (record r belongs to LUT entry idx with
	// lut[idx] <= r < lut[idx+1]; its prefix is idx & prefix_mask), and a parallel memcpy of a record range.
	uint64_t records() const { return avail_; }
	uint32_t record_bytes() const { return rec_bytes_; }
	uint32_t suffix_bytes() const { return suf_bytes_; }
	uint32_t counter_bytes() const { return counter_size_; }
	uint64_t prefix_mask() const { return prefix_mask_; }
	const LutVec &lut() const { lut_loaded(); return lut_; }     // host copy, loaded on first use (empty + io_failed() if that fails)
	size_t lut_entries() const { return n_lut_ ? n_lut_ + 1 : 0; }   // with the sentinel
	bool read_lut(uint64_t *dst) const;                          // lut_entries() words straight from the file (parallel preads)
	void copy_records(uint64_t rec0, uint64_t n, unsigned char *dst) const;

private:
	size_t decode_range(const unsigned char *recs, uint64_t rec0, size_t n_recs, uint64_t *kmers, uint32_t *counts) const;
	int threads_ = 1;
	// The record file is read with pread (page cache -> destination, no mapping: on a 0.8 GB file the first-touch
	// faults of a fresh mapping cost pass 1 ~6 ms and its munmap another ~12 ms); records start at byte 4.
	int fd_ = -1;
	size_t file_len_ = 0;
	bool read_at(uint64_t off, void *dst, size_t bytes) const;
	mutable std::atomic<bool> io_failed_{false};
	std::vector<unsigned char> stage_;       // raw bytes of the batch next_batch is decoding (host decoder only)
	mutable LutVec lut_;             // concatenated LUT(s); lut_[size] sentinel = total
	bool lut_loaded() const;
	int pre_fd_ = -1;
	size_t n_lut_ = 0;
	uint64_t rec_ = 0, avail_ = 0, total_ = 0, max_count_ = 0, prefix_mask_ = 0;
	uint32_t k_ = 0, mode_ = 0, counter_size_ = 0, p_ = 0, min_count_ = 0, version_ = 0;
	uint32_t suf_bytes_ = 0, rec_bytes_ = 0;
	std::string err_;
};

}   // namespace kmx
