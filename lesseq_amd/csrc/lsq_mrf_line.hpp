// The MRF_SINGLE line splitter, shared by the host parser (lsq_mrf.cpp) and the device parser
// (lsq_ingest.hip): one source for the reference's field arithmetic (count/count.cpp:297-326).
//
// The reference walks a line with std::string::find / substr.  Two of its habits matter for
// odd lines and are kept: `npos + 1 == 0` (a search that starts "after" a colon that was not
// found restarts at the beginning of the line) and substr lengths computed from npos (the
// field then runs to the end of the line).  Fields 3 and 4 of every comma-separated block go
// through boost::lexical_cast<long>: the whole field, optional sign, decimal digits, no
// overflow -- anything else ends the program with status 1.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define LSQ_HD __host__ __device__
#else
#define LSQ_HD
#endif

namespace lsq {

constexpr size_t MRF_NPOS = (size_t)-1;

struct MrfView { const char *p; size_t n; };

LSQ_HD inline size_t mrf_find(MrfView l, char c, size_t pos) {
	for (size_t i = pos; i < l.n; ++i) if (l.p[i] == c) return i;
	return MRF_NPOS;
}
LSQ_HD inline MrfView mrf_sub(MrfView l, size_t pos, size_t cnt) {
	if (pos > l.n) pos = l.n;
	const size_t avail = l.n - pos;
	return MrfView{l.p + pos, cnt < avail ? cnt : avail};
}
LSQ_HD inline bool mrf_cast_long(MrfView f, int64_t &out) {
	if (f.n == 0) return false;
	const size_t i0 = (f.p[0] == '+' || f.p[0] == '-') ? 1 : 0;
	if (i0 == f.n) return false;
	uint64_t v = 0;
	for (size_t j = i0; j < f.n; ++j) {
		const unsigned d = (unsigned)(unsigned char)f.p[j] - (unsigned)'0';
		if (d > 9) return false;
		if (v > (UINT64_MAX - d) / 10) return false;
		v = v * 10 + d;
	}
	if (f.p[0] == '-') { if (v > (uint64_t)INT64_MAX + 1) return false; out = (int64_t)(0 - v); }
	else { if (v > (uint64_t)INT64_MAX) return false; out = (int64_t)v; }
	return true;
}

// lines that consume a line number but make no read (count/count.cpp:288)
LSQ_HD inline bool mrf_line_is_skipped(MrfView l) {
	if (l.n >= 1 && l.p[0] == '#') return true;
	if (l.n != 15) return false;
	const char *k = "AlignmentBlocks";
	for (int i = 0; i < 15; ++i) if (l.p[i] != k[i]) return false;
	return true;
}

// Calls on_block(chr, strand, start, end) for every block of the line, in order.  Returns false
// at the first field that fails the cast (blocks before it have been delivered).
template <class OnBlock>
LSQ_HD inline bool mrf_split_line(MrfView line, OnBlock &&on_block) {
	size_t last_comma = 0;
	while (last_comma != MRF_NPOS) {
		size_t colon = mrf_find(line, ':', last_comma);
		const size_t cpos = last_comma == 0 ? 0 : last_comma + 1;
		const MrfView chr = mrf_sub(line, cpos, colon - cpos);
		size_t old_colon = colon;
		colon = mrf_find(line, ':', colon + 1);
		const MrfView strand = mrf_sub(line, old_colon + 1, colon - old_colon - 1);
		old_colon = colon;
		colon = mrf_find(line, ':', colon + 1);
		int64_t start, end;
		if (!mrf_cast_long(mrf_sub(line, old_colon + 1, colon - old_colon - 1), start)) return false;
		old_colon = colon;
		colon = mrf_find(line, ':', colon + 1);
		if (!mrf_cast_long(mrf_sub(line, old_colon + 1, colon - old_colon - 1), end)) return false;
		on_block(chr, strand, start, end);
		last_comma = mrf_find(line, ',', colon);
	}
	return true;
}

} // namespace lsq
