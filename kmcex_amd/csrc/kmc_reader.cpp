// kmc_reader.cpp -- see kmc_reader.h.  Format per SURVEY.md Appendix B.3 (kmc_file.cpp:132-292, :428-515).
#include "kmc_reader.h"
#include <cstring>

namespace kmx {

namespace {
constexpr size_t kChunk = size_t(1) << 25;            // same read granularity as the reference (kmc_file.cpp:18)

bool read_file(const std::string &path, std::vector<unsigned char> &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	rewind(f);
	out.resize(sz > 0 ? (size_t)sz : 0);
	size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
	fclose(f);
	return got == out.size();
}
template <typename T> T rd(const unsigned char *p)
{
	T v;
	memcpy(&v, p, sizeof(T));
	return v;
}
}   // namespace

bool KmcListing::open(const std::string &prefix)
{
	close();
	std::vector<unsigned char> pre;
	if (!read_file(prefix + ".kmc_pre", pre) || pre.size() < 4 + 4 + 12 || memcmp(pre.data(), "KMCP", 4) ||
	    memcmp(pre.data() + pre.size() - 4, "KMCP", 4)) {
		err_ = "cannot open " + prefix + ".kmc_pre (missing or no KMCP markers)";
		return false;
	}
	const size_t end = pre.size();
	version_ = rd<uint32_t>(&pre[end - 12]);                        // kmc_file.cpp:184-187
	const uint32_t header_offset = pre[end - 8];                    // low byte only (:193, :243)
	if (version_ != 0 && version_ != 0x200) { err_ = "unsupported KMC database version"; return false; }
	if ((size_t)header_offset + 8 > end - 4) { err_ = "corrupt KMC header offset"; return false; }
	const unsigned char *h = &pre[end - 8 - header_offset];
	size_t lut_bytes;
	if (version_ == 0x200) {                                        // KMC2 header (:197-208)
		if (header_offset < 37) { err_ = "corrupt KMC2 header"; return false; }
		k_ = rd<uint32_t>(h);
		mode_ = rd<uint32_t>(h + 4);
		counter_size_ = rd<uint32_t>(h + 8);
		p_ = rd<uint32_t>(h + 12);
		const uint32_t sig_len = rd<uint32_t>(h + 16);
		min_count_ = rd<uint32_t>(h + 20);
		max_count_ = rd<uint32_t>(h + 24);
		total_ = rd<uint64_t>(h + 28);
		const size_t sig_map_bytes = ((size_t(1) << (2 * sig_len)) + 1) * 4;
		const size_t inner = end - 8 - 4;                           // without both markers and the offset field
		if (inner < sig_map_bytes + header_offset + 8) { err_ = "corrupt KMC2 prefix file"; return false; }
		lut_bytes = inner - (sig_map_bytes + header_offset + 8);    // (:212); one guard u64 follows the LUTs
	} else {                                                        // KMC1 header (:253-279)
		if (header_offset < 40) { err_ = "corrupt KMC1 header"; return false; }
		k_ = rd<uint32_t>(h);
		mode_ = rd<uint32_t>(h + 4);
		counter_size_ = rd<uint32_t>(h + 8);
		p_ = rd<uint32_t>(h + 12);
		min_count_ = rd<uint32_t>(h + 16);
		max_count_ = rd<uint32_t>(h + 20);
		total_ = rd<uint64_t>(h + 24);
		max_count_ += rd<uint64_t>(h + 32) & 0xFFFFFFFF00000000ULL;
		lut_bytes = end - 8 - 4 - header_offset;
	}
	if (mode_ != 0) { err_ = "KMC databases with quality-aware (float) counters are not supported"; return false; }
	if (k_ == 0 || k_ > 64 || p_ > 15 || p_ >= k_ || (k_ - p_) % 4 || counter_size_ == 0 || counter_size_ > 4) {
		err_ = "unsupported KMC parameters (k, prefix length or counter size)";
		return false;
	}
	const size_t n_lut = lut_bytes / 8;
	if (n_lut == 0 || 4 + n_lut * 8 > end) { err_ = "corrupt KMC LUT"; return false; }
	lut_.resize(n_lut + 1);
	memcpy(lut_.data(), &pre[4], n_lut * 8);
	lut_[n_lut] = total_;
	prefix_mask_ = (uint64_t(1) << (2 * p_)) - 1;
	suf_bytes_ = (k_ - p_) / 4;
	rec_bytes_ = suf_bytes_ + counter_size_;
	suf_ = fopen((prefix + ".kmc_suf").c_str(), "rb");
	char mk[4];
	if (!suf_ || fread(mk, 1, 4, suf_) != 4 || memcmp(mk, "KMCS", 4)) {
		err_ = "cannot open " + prefix + ".kmc_suf (missing or no KMCS marker)";
		close();
		return false;
	}
	buf_.resize(kChunk - kChunk % rec_bytes_);
	restart();
	return true;
}

void KmcListing::close()
{
	if (suf_) fclose(suf_);
	suf_ = nullptr;
	lut_.clear();
}

void KmcListing::restart()
{
	if (!suf_) return;
	fseek(suf_, 4, SEEK_SET);
	buf_pos_ = buf_len_ = 0;
	lut_idx_ = 0;
	rec_ = 0;
}

bool KmcListing::fill()
{
	buf_len_ = fread(buf_.data(), 1, buf_.size(), suf_);
	buf_pos_ = 0;
	return buf_len_ >= rec_bytes_;
}

size_t KmcListing::next_batch(uint64_t *kmers, uint32_t *counts, size_t max_n)
{
	const int W = words();
	const int sbits = 8 * (int)suf_bytes_;
	const size_t n_lut = lut_.size() - 1;
	size_t out = 0;
	while (out < max_n && rec_ < total_) {
		if (buf_pos_ + rec_bytes_ > buf_len_ && !fill()) break;      // truncated file: stop like EOF
		while (lut_idx_ + 1 < n_lut && lut_[lut_idx_ + 1] <= rec_) lut_idx_++;   // (:439-445) skip empty prefixes
		const unsigned char *r = &buf_[buf_pos_];
		buf_pos_ += rec_bytes_;
		rec_++;
		uint32_t c = 0;
		for (uint32_t b = 0; b < counter_size_; b++) c |= (uint32_t)r[suf_bytes_ + b] << (8 * b);
		if (c < min_count_ || c > max_count_) continue;                 // (:513)
		unsigned __int128 v = lut_idx_ & prefix_mask_;                  // (:449) "& prefix_mask" for KMC2
		for (uint32_t b = 0; b < suf_bytes_; b++) v = (v << 8) | r[b];
		(void)sbits;
		if (W == 1) kmers[out] = (uint64_t)v;
		else { kmers[2 * out] = (uint64_t)(v >> 64); kmers[2 * out + 1] = (uint64_t)v; }
		counts[out] = c;
		out++;
	}
	return out;
}

}   // namespace kmx
