// variants_host.h -- host side of SURVEY.md §8f row f2: what callsemble does with an event after the tally
// (src/indelope.nim:375-428), the last-two-variants dedupe of the main loop (:604-608) and `$`(Variant) (:104-113,
// src/genotyper.nim:31-34).  Pure host code over one batch's inputs and flat results: string slicing, a handful of
// fp64 divisions, a median.  No kernel: the reference does this per printed variant, a few thousand times per
// chromosome.
//
// AKE/RKE and the filter of :412 need the `kmer` package's distance `d` (indelope.nimble:10-11; un-vendored, no
// version pinned): taken as the distance of the k-mer window from the closer end of the read.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#include "indelope_hip.h"

namespace ihp_host {

// the ops Ez.cigar yields (ksw2.nim:22-33): the full CIGAR cut before the op at which max_q query bases are consumed
struct EzCigar {
	const uint32_t *w; int n;
	EzCigar(const uint32_t *words, int n_cigar, int max_q) : w(words), n(0)
	{
		const uint32_t stop = (uint32_t)max_q;
		uint32_t used = 0;
		while (n < n_cigar && used < stop) {
			if (op(n) != 2) used += len(n);
			++n;
		}
	}
	int op(int i) const { return (int)(w[i] & 0xf); }
	uint32_t len(int i) const { return w[i] >> 4; }
	std::string str() const                               // cigar_string, ksw2.nim:41-49
	{
		std::string s;
		for (int i = 0; i < n; ++i) { s += std::to_string(len(i)); s += "MID"[op(i)]; }
		return s;
	}
	// get_min_flank, indelope.nim:119-132: the M run before the event and the one after it, whichever is shorter
	long long min_flank(int event_type, uint32_t event_len) const
	{
		const long long unset = std::numeric_limits<long long>::max();
		long long flank = unset;
		bool seen = false;
		for (int i = 0; i < n; ++i) {
			if (op(i) == 0) {
				flank = seen ? std::min<long long>(len(i), flank) : (long long)len(i);
				if (seen) return flank;
			} else if (op(i) - 1 == event_type && len(i) == event_len) {
				if (flank == unset) flank = 0;
				seen = true;
			}
		}
		return 0;
	}
};

inline bool one_letter(const char *s, int n) { return std::all_of(s, s + n, [&](char c) { return c == s[0]; }); }

struct Emitted { long long start; std::string ref, alt; };

inline int call_variants(const ihp_params &P, const ihp_batch_in &in, const ihp_batch_out &out, ihp_variants *vars)
{
	std::vector<ihp_variant> recs;
	std::string pool;
	auto put = [&](const char *s, long long n) { const long long at = (long long)pool.size(); pool.append(s, (size_t)n); return at; };
	std::vector<Emitted> printed;                          // last_var, last_var2 (:598-599, :604-608)
	std::vector<uint8_t> quals_of_hits;
	const int K = P.K;
	for (int32_t r = 0; r < out.n_regions; ++r) {
		const long long first_read = in.region_read_off[r], depth = in.region_read_off[r + 1] - first_read;
		const char *ref = (const char *)in.ref_bases + in.ref_off[r];
		const long long ref_len = in.ref_off[r + 1] - in.ref_off[r], ref_at = in.ref_origin[r];
		for (long long c = out.contig_off[r]; c < out.contig_off[r + 1]; ++c) {
			const EzCigar cig(out.cigar + out.cigar_off[c], out.aln_ez[c].n_cigar, out.aln_ez[c].max_q);
			const long long contig_len = out.ctg_seq_off[c + 1] - out.ctg_seq_off[c];
			// (IHP_FETCH_COMPACT: 4-bit bases, expanded for the contigs an insertion's allele is cut from -- below -- and no others)
			std::string contig_buf;
			const char *contig = out.ctg_seq ? (const char *)out.ctg_seq + out.ctg_seq_off[c] : nullptr;
			auto contig_bases = [&]() -> const char * {
				if (contig) return contig;
				if (contig_buf.empty() && contig_len > 0) {
					contig_buf.resize((size_t)contig_len);
					const uint8_t *p4 = out.ctg_seq4 + (out.ctg_seq_off[c] >> 1) + c;
					for (long long i = 0; i < contig_len; ++i) contig_buf[(size_t)i] = "=ACMGRSVTWYHKDBN"[(p4[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
				}
				return contig_buf.data();
			};
			for (long long e = out.event_off[c]; e < out.event_off[c + 1]; ++e) {
				const ihp_event &E = out.events[e];
				if (E.status != IHP_EV_TALLIED) continue;
				ihp_variant v;
				memset(&v, 0, sizeof(v));
				v.region = r; v.contig = (int32_t)(c - out.contig_off[r]); v.event = e;
				v.start = E.tstart; v.gt = E.gt; v.gq = v.qual = E.qual;
				std::copy(E.gl, E.gl + 3, v.gl);
				v.ad[0] = E.ref_support; v.ad[1] = E.alt_support; v.event_type = E.type;
				v.amq = v.rmq = -1; v.ake = v.rke = std::nan("");
				memcpy(v.ref_kmer, E.ref_kmer, sizeof(v.ref_kmer)); memcpy(v.alt_kmer, E.alt_kmer, sizeof(v.alt_kmer));
				// the reference's `continue`s, in its order; the first one that fires is the verdict
				auto verdict = [&]() -> int {
					if (E.alt_support < P.min_reads) return IHP_VF_LOW_ALT;                              // :375
					if ((double)E.alt_support / (double)depth < 0.1) return IHP_VF_LOW_FRAC;             // :377
					if (E.gt == IHP_GT_HOM_REF) return IHP_VF_HOM_REF;                                   // :380
					if (E.cf_offset == 0 && E.both_found >= (int)(0.75 * (double)std::min(E.ref_support, E.alt_support)))
						return IHP_VF_BOTH_AT_EDGE;                                                      // :384
					v.dp = (int32_t)depth;                                                               // :386
					if (E.cf_offset < 5) { v.lo = 1; v.qual /= 2.0; }                                    // :387-389
					if (E.both_found > 0) { v.bs = E.both_found; v.qual /= 1.5; } else v.qual *= 2;      // :390-394
					const std::string cc = cig.str();                                                    // :395
					v.cc_off = put(cc.data(), (long long)cc.size()); v.cc_len = (int32_t)cc.size();
					v.al = E.aligned;                                                                    // :396-397
					const long long flank = cig.min_flank(E.type, E.len);                                // :398
					if (flank - 1 < std::max(E.tstop - E.tstart, E.qstop - E.qstart)) return IHP_VF_SMALL_FLANK;   // :400
					v.mf = (int32_t)flank; v.cf = E.cf_offset; v.nc = out.n_contigs_pre[r];              // :401-403
					if (E.cf_offset == 0) v.qual /= 4.0;                                                 // :404-405
					// rdists/adists/rmapqs/amapqs of :302-309, rebuilt from the first-hit windows
					for (int alt = 0; alt < 2; ++alt) {
						const int32_t *hit = (alt ? out.alt_hit : out.ref_hit) + out.hit_off[e];
						double sum = 0; long long cnt = 0;
						quals_of_hits.clear();
						for (long long i = 0; i < depth; ++i) {
							if (hit[i] < 0) continue;
							const long long L = in.read_off[first_read + i + 1] - in.read_off[first_read + i];
							sum += (double)std::min<long long>(hit[i], L - K - hit[i]);
							++cnt;
							quals_of_hits.push_back(in.mapq[first_read + i]);
						}
						(alt ? v.ake : v.rke) = sum / (double)cnt;                                       // mean(), :146-150 (0/0 when empty)
						if (!quals_of_hits.empty()) {                                                    // median(), :152-155
							std::sort(quals_of_hits.begin(), quals_of_hits.end());
							(alt ? v.amq : v.rmq) = quals_of_hits[(size_t)((double)quals_of_hits.size() / 2.0)];
						}
					}
					if (v.ake < 5) return IHP_VF_KMER_AT_END;                                            // :412
					std::string ref_allele, alt_allele;
					if (E.type == 1) {                                                                   // deletion, :413-415
						const long long lo = E.tstart - 1 - ref_at, hi = E.tstop - 1 - ref_at;          // fai.get is end-inclusive
						if (lo < 0 || hi >= ref_len || hi < lo) return IHP_VF_OOB;
						ref_allele.assign(ref + lo, (size_t)(hi - lo + 1));
						alt_allele.assign(ref_allele, 0, 1);
					} else {                                                                             // insertion, :420-427
						const long long at = E.tstart - 1 - ref_at;
						if (at < 0 || at >= ref_len || E.qstart < 1 || E.qstop > contig_len) return IHP_VF_OOB;
						ref_allele.assign(ref + at, 1);
						alt_allele.assign(contig_bases() + E.qstart - 1, (size_t)(E.qstop - E.qstart + 1));
					}
					v.ref_off = put(ref_allele.data(), (long long)ref_allele.size()); v.ref_len = (int32_t)ref_allele.size();
					v.alt_off = put(alt_allele.data(), (long long)alt_allele.size()); v.alt_len = (int32_t)alt_allele.size();
					if (E.type == 0 && K >= 11 && one_letter(alt_allele.data() + 1, (int)alt_allele.size() - 1) &&
					    one_letter(E.alt_kmer + K - 11, 11) && one_letter(E.ref_kmer + K - 11, 11))
						return IHP_VF_HOMOPOLYMER;                                                       // :423-427
					for (const Emitted &o : printed)                                                     // :604-605
						if (o.start == v.start && o.ref == ref_allele && o.alt == alt_allele) return IHP_VF_DUPLICATE;
					printed.insert(printed.begin(), Emitted{v.start, ref_allele, alt_allele});           // :607-608
					if (printed.size() > 2) printed.pop_back();
					return IHP_VF_EMITTED;
				};
				v.filter = verdict();
				recs.push_back(v);
			}
		}
	}
	vars->n = (int64_t)recs.size();
	vars->v = (ihp_variant *)calloc(recs.size() ? recs.size() : 1, sizeof(ihp_variant));
	vars->n_chars = (int64_t)pool.size();
	vars->chars = (char *)calloc(pool.size() + 1, 1);
	if (!vars->v || !vars->chars) { free(vars->v); free(vars->chars); memset(vars, 0, sizeof(*vars)); return IHP_E_NOMEM; }
	if (!recs.empty()) memcpy(vars->v, recs.data(), recs.size() * sizeof(ihp_variant));
	memcpy(vars->chars, pool.data(), pool.size());
	return 0;
}

// Nim's formatFloat(x, ffDecimal, precision)
inline std::string decimal(double x, int precision)
{
	if (std::isnan(x)) return "nan";
	if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
	char t[64];
	snprintf(t, sizeof(t), "%.*f", precision, x);
	return t;
}

inline std::string vcf_line(const ihp_variant &v, const char *chars, const char *chrom)
{
	static const char *const gts[4] = {"0/0", "0/1", "1/1", "./."};      // genotyper.nim:17
	std::string info = "AD=" + std::to_string(v.ad[0]) + "," + std::to_string(v.ad[1]) + ";ref_kmer=" + v.ref_kmer +
	                   ";alt_kmer=" + v.alt_kmer;                         // info(), :62-68
	info += ";DP=" + std::to_string(v.dp);                                // the info_add calls of :386-411, in order
	if (v.lo) info += ";LO";
	if (v.bs > 0) info += ";BS=" + std::to_string(v.bs);
	info += ";CC=" + std::string(chars + v.cc_off, (size_t)v.cc_len);
	if (v.al) info += ";AL";
	info += ";MF=" + std::to_string(v.mf) + ";CF=" + std::to_string(v.cf) + ";NC=" + std::to_string(v.nc);
	info += ";AKE=" + decimal(v.ake, 2) + ";RKE=" + decimal(v.rke, 2);
	if (v.amq >= 0) info += ";AMQ=" + std::to_string(v.amq);
	if (v.rmq >= 0) info += ";RMQ=" + std::to_string(v.rmq);
	const std::string sample = std::string(gts[v.gt & 3]) + ":" + decimal(v.gq, 4) + ":" + decimal(v.gl[0], 4) + "," +
	                           decimal(v.gl[1], 4) + "," + decimal(v.gl[2], 4);   // `$`(Genotype), genotyper.nim:31-34
	return std::string(chrom) + "\t" + std::to_string(v.start) + "\t.\t" + std::string(chars + v.ref_off, (size_t)v.ref_len) + "\t" +
	       std::string(chars + v.alt_off, (size_t)v.alt_len) + "\t" + decimal(v.qual, 2) + "\tPASS\t" + info + "\tGT:GQ:GL\t" + sample;
}

}  // namespace ihp_host
