// The MRF_SINGLE line splitter, shared by the host parser (lsq_mrf.cpp) and the device parser
// (lsq_ingest.hip): one source for the reference's field arithmetic (count/count.cpp:297-326).
//
// The reference walks a line with std::string::find / substr.  Two of its habits matter for
// odd lines and are kept: `npos + 1 == 0` (a search that starts "after" a colon that was not
// found restarts at the beginning of the line) and substr lengths computed from npos (the
// field then runs to the end of the line).  Fields 3 and 4 of every comma-separated block go
// through boost::lexical_cast<long>: the whole field, optional sign, decimal digits, no
// overflow -- anything else ends the program with status 1.
//
// The splitter is a template over the view of the line: the host parser and the device's
// fall-back for very long lines walk plain memory with size_t positions; the device parser
// proper walks a tile of the text in LDS (an address-space-3 pointer: ds_read_u8, not a flat
// load) with 32-bit positions.  The arithmetic is the same modulo 2^32 and modulo 2^64 for any
// line shorter than 2^31 bytes: every difference the reference forms is either a true length,
// or taken from npos and then cut to the bytes that are left (mrf_sub).
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

template <class Ptr, class Idx>
struct MrfViewT {
	Ptr p; Idx n;
	typedef Idx index_type;
	static constexpr Idx npos = (Idx)-1;
};
typedef MrfViewT<const char *, size_t> MrfView;

template <class V>
LSQ_HD inline typename V::index_type mrf_find(V l, char c, typename V::index_type pos) {
	for (typename V::index_type i = pos; i < l.n; ++i) if (l.p[i] == c) return i;
	return V::npos;
}
template <class V>
LSQ_HD inline V mrf_sub(V l, typename V::index_type pos, typename V::index_type cnt) {
	if (pos > l.n) pos = l.n;
	const typename V::index_type avail = l.n - pos;
	return V{l.p + pos, cnt < avail ? cnt : avail};
}
template <class V>
LSQ_HD inline bool mrf_cast_long(V f, int64_t &out) {
	typedef typename V::index_type Idx;
	if (f.n == 0) return false;
	const char c0 = f.p[0];
	const Idx i0 = (c0 == '+' || c0 == '-') ? 1 : 0;
	if (i0 == f.n) return false;
	uint64_t v = 0;
	if (f.n - i0 <= 9) {
		// at most nine digits: the value fits 32 bits and nothing can overflow (every coordinate of a real file)
		uint32_t w = 0;
		for (Idx j = i0; j < f.n; ++j) {
			const unsigned d = (unsigned)(unsigned char)f.p[j] - (unsigned)'0';
			if (d > 9) return false;
			w = w * 10u + d;
		}
		v = w;
	} else {
		for (Idx j = i0; j < f.n; ++j) {
			const unsigned d = (unsigned)(unsigned char)f.p[j] - (unsigned)'0';
			if (d > 9) return false;
			// v * 10 + d <= UINT64_MAX  <=>  v <= (UINT64_MAX - d) / 10; UINT64_MAX / 10 = 1844674407370955161 rest 5
			if (v > 1844674407370955161ull || (v == 1844674407370955161ull && d > 5)) return false;
			v = v * 10 + d;
		}
	}
	if (c0 == '-') { if (v > (uint64_t)INT64_MAX + 1) return false; out = (int64_t)(0 - v); }
	else { if (v > (uint64_t)INT64_MAX) return false; out = (int64_t)v; }
	return true;
}

// lines that consume a line number but make no read (count/count.cpp:288)
template <class V>
LSQ_HD inline bool mrf_line_is_skipped(V l) {
	if (l.n >= 1 && l.p[0] == '#') return true;
	if (l.n != 15) return false;
	const char *k = "AlignmentBlocks";
	for (int i = 0; i < 15; ++i) if (l.p[i] != k[i]) return false;
	return true;
}

// Calls on_block(chr, strand, start, end) for every block of the line, in order.  Returns false
// at the first field that fails the cast (blocks before it have been delivered).
template <class V, class OnBlock>
LSQ_HD inline bool mrf_split_line(V line, OnBlock &&on_block) {
	typedef typename V::index_type Idx;
	Idx last_comma = 0;
	while (last_comma != V::npos) {
		Idx colon = mrf_find(line, ':', last_comma);
		const Idx cpos = last_comma == 0 ? 0 : last_comma + 1;
		const V chr = mrf_sub(line, cpos, colon - cpos);
		Idx old_colon = colon;
		colon = mrf_find(line, ':', colon + 1);
		const V strand = mrf_sub(line, old_colon + 1, colon - old_colon - 1);
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
