// synth.cpp -- deterministic synthetic candidate regions.
//
// Host-only helper (g++, no HIP): builds the flat `ihp_batch_in` arrays that the
// BAM sweep would otherwise produce, so tests and bench.py see identical inputs in
// the build container and on the GPU box.  Every region is generated from its own
// PRNG stream (seed, region index), so a rank can generate just its shard.
//
// Per region: uniform-ACGT reference window W of length L; `n_events` planted
// indels (50/50 insertion/deletion, length U[5,40]) define the alt haplotype; n
// reads of `read_len` drawn 50/50 from the two haplotypes, starts uniform so that
// every read spans the (first) event with >= 15 bp on both sides, optional
// substitution errors, then sorted by start (stable) = BAM order.  Read start/stop
// are mapped back to W coordinates the way an aligner would report them.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sched.h>

extern "C" {

typedef struct {
	uint64_t seed;
	int32_t  n_regions;
	int32_t  first_region;     // global index of region 0 of this shard
	int32_t  read_len;
	int32_t  n_reads_min, n_reads_max;   // equal: fixed; else log-uniform
	int32_t  n_events;         // 1, or 2 (second event 250 bp right of the first)
	int32_t  window_len;       // 0: 2*read_len + 200
	int32_t  event_pos;        // 0: window_len / 2
	double   err_rate;
	int64_t  origin0;          // genomic origin of region 0's window
	int64_t  origin_step;
	double   dup_frac;         // fraction of events planted as tandem duplications (an insertion of >= 24 bp
	                           // that repeats the reference bases to its right): the alt k-mer then also
	                           // occurs in the reference haplotype, reads carry both k-mers and the
	                           // alignment fallback of indelope.nim:312-372 has to decide
} ihp_synth_cfg;

struct Rng {
	uint64_t s;
	explicit Rng(uint64_t seed) {
		// splitmix64 to decorrelate neighbouring (seed, region) pairs
		uint64_t z = seed + 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		s = z ^ (z >> 31);
		if (!s) s = 0x1DE10BEull;
	}
	uint64_t next() {            // xorshift64*
		s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
		return s * 0x2545F4914F6CDD1Dull;
	}
	uint32_t below(uint32_t n) { return (uint32_t)((next() >> 11) % n); }
	double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

static int window_len(const ihp_synth_cfg *c) { return c->window_len ? c->window_len : 2 * c->read_len + 200; }

static int region_nreads(const ihp_synth_cfg *c, Rng &g)
{
	if (c->n_reads_min >= c->n_reads_max) return c->n_reads_min;
	double lo = std::log((double)c->n_reads_min), hi = std::log((double)c->n_reads_max + 1.0);
	int n = (int)std::exp(lo + (hi - lo) * g.unit());
	return std::min(std::max(n, c->n_reads_min), c->n_reads_max);
}

static uint64_t region_seed(const ihp_synth_cfg *c, int r)
{
	return c->seed * 0xD1342543DE82EF95ull + (uint64_t)(c->first_region + r) * 0x9E3779B97F4A7C15ull;
}

// sizes: n_reads, n_bases, n_ref
int ihp_synth_sizes(const ihp_synth_cfg *c, int64_t *n_reads, int64_t *n_bases, int64_t *n_ref)
{
	int64_t nr = 0;
	for (int r = 0; r < c->n_regions; ++r) {
		Rng g(region_seed(c, r));
		nr += region_nreads(c, g);
	}
	*n_reads = nr; *n_bases = nr * c->read_len; *n_ref = (int64_t)c->n_regions * window_len(c);
	return 0;
}

// regions [r_lo, r_hi); ri = index of region r_lo's first read (regions are independent PRNG streams)
static void fill_range(const ihp_synth_cfg *c, int r_lo, int r_hi, int64_t ri, int64_t *region_read_off, int64_t *read_off, uint8_t *bases,
                       uint8_t *quals, int64_t *read_start, int64_t *read_stop, uint8_t *mapq, uint8_t *read_skip,
                       int64_t *ref_off, uint8_t *ref_bases, int64_t *ref_origin, int32_t *truth)
{
	static const char ACGT[] = "ACGT";
	const int L = window_len(c), RL = c->read_len;
	std::vector<uint8_t> alt; std::vector<int32_t> amap;
	struct Rd { int hap; int s; int order; };
	std::vector<Rd> rds; std::vector<uint8_t> tmp;
	for (int r = r_lo; r < r_hi; ++r) {
		Rng g(region_seed(c, r));
		const int n = region_nreads(c, g);
		uint8_t *W = ref_bases + (int64_t)r * L;
		for (int i = 0; i < L; ++i) W[i] = (uint8_t)ACGT[g.below(4)];
		const int64_t origin = c->origin0 + (int64_t)(c->first_region + r) * c->origin_step;
		ref_origin[r] = origin; ref_off[r + 1] = (int64_t)(r + 1) * L;
		// alt haplotype + map alt position -> W position
		const int p0 = c->event_pos ? c->event_pos : L / 2;
		int ev_pos[2] = {p0, p0 + 250}, ev_type[2] = {-1, -1}, ev_len[2] = {0, 0};
		int ev_dup[2] = {0, 0};
		for (int e = 0; e < c->n_events && e < 2; ++e) {
			ev_type[e] = (int)g.below(2); ev_len[e] = 5 + (int)g.below(36);
			if (c->dup_frac > 0 && g.unit() < c->dup_frac) { ev_dup[e] = 1; ev_type[e] = 0; ev_len[e] = 24 + (int)g.below(17); }
		}
		alt.clear(); amap.clear();
		int w = 0;
		for (int e = 0; e < 2; ++e) {
			if (ev_type[e] < 0) continue;
			for (; w < ev_pos[e]; ++w) { alt.push_back(W[w]); amap.push_back(w); }
			if (ev_type[e] == 1) w += ev_len[e];                       // deletion: skip W bases
			else if (ev_dup[e]) for (int k = 0; k < ev_len[e]; ++k) { alt.push_back(W[w + k]); amap.push_back(w); }
			else for (int k = 0; k < ev_len[e]; ++k) { alt.push_back((uint8_t)ACGT[g.below(4)]); amap.push_back(w); }
		}
		for (; w < L; ++w) { alt.push_back(W[w]); amap.push_back(w); }
		if (truth) { truth[r * 4] = ev_type[0]; truth[r * 4 + 1] = ev_len[0]; truth[r * 4 + 2] = ev_type[1]; truth[r * 4 + 3] = ev_len[1]; }
		// reads
		int slo, shi;
		if (c->n_events >= 2) { slo = p0 - (RL - 15); shi = p0 + 235; }
		else { slo = p0 - RL + 15; shi = p0 - 15; }
		if (slo < 0) slo = 0;
		if (shi < slo) shi = slo;                                  // reads shorter than 30 bases: all start at the same place
		if (shi > L - RL) shi = L - RL > slo ? L - RL : slo;
		rds.resize(n);
		for (int i = 0; i < n; ++i) {
			rds[i].hap = (int)g.below(2);
			int hl = rds[i].hap ? (int)alt.size() : L;
			int s = slo + (int)g.below((uint32_t)(shi - slo + 1));
			if (s + RL > hl) s = hl - RL;
			if (s < 0) s = 0;
			rds[i].s = s; rds[i].order = i;
		}
		// error draws happen in generation order so that sorting does not change the stream
		tmp.resize((size_t)n * RL);
		for (int i = 0; i < n; ++i) {
			const uint8_t *h = rds[i].hap ? alt.data() : W;
			uint8_t *o = tmp.data() + (size_t)i * RL;
			for (int k = 0; k < RL; ++k) {
				uint8_t b = h[rds[i].s + k];
				if (c->err_rate > 0 && g.unit() < c->err_rate) {
					uint8_t nb;
					do nb = (uint8_t)ACGT[g.below(4)]; while (nb == b);
					b = nb;
				}
				o[k] = b;
			}
		}
		auto wstart = [&](const Rd &x) { return x.hap ? amap[x.s] : x.s; };
		std::stable_sort(rds.begin(), rds.end(), [&](const Rd &a, const Rd &b) { return wstart(a) < wstart(b); });
		for (int i = 0; i < n; ++i, ++ri) {
			const Rd &x = rds[i];
			memcpy(bases + ri * RL, tmp.data() + (size_t)x.order * RL, (size_t)RL);
			memset(quals + ri * RL, 30, (size_t)RL);
			int ws = wstart(x), we = x.hap ? amap[x.s + RL - 1] + 1 : x.s + RL;
			read_start[ri] = origin + ws; read_stop[ri] = origin + we;
			mapq[ri] = 60; read_skip[ri] = 0;
			read_off[ri + 1] = (ri + 1) * RL;
		}
		region_read_off[r + 1] = ri;
	}
}

int ihp_synth_fill(const ihp_synth_cfg *c, int64_t *region_read_off, int64_t *read_off, uint8_t *bases,
                   uint8_t *quals, int64_t *read_start, int64_t *read_stop, uint8_t *mapq, uint8_t *read_skip,
                   int64_t *ref_off, uint8_t *ref_bases, int64_t *ref_origin,
                   int32_t *truth /* [n_regions*4]: type0,len0,type1,len1 (type 0 ins, 1 del; -1 none) */)
{
	region_read_off[0] = 0; read_off[0] = 0; ref_off[0] = 0;
	const int R = c->n_regions;
	// threads over contiguous region ranges (the CPUs this process may use); the output does not depend on the count
	int nt = 1;
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) nt = CPU_COUNT(&set);
	if (const char *e = getenv("IHP_SYNTH_THREADS")) nt = atoi(e);
	if (nt > 64) nt = 64;
	if (nt > R / 256) nt = R / 256;
	if (nt < 1) nt = 1;
	std::vector<int64_t> first((size_t)nt + 1, 0);
	for (int t = 0; t < nt; ++t) {
		int64_t n = 0;
		for (int r = (int)((int64_t)R * t / nt); r < (int)((int64_t)R * (t + 1) / nt); ++r) { Rng g(region_seed(c, r)); n += region_nreads(c, g); }
		first[(size_t)t + 1] = first[(size_t)t] + n;
	}
	std::vector<std::thread> th;
	for (int t = 0; t < nt; ++t)
		th.emplace_back(fill_range, c, (int)((int64_t)R * t / nt), (int)((int64_t)R * (t + 1) / nt), first[(size_t)t], region_read_off, read_off,
		                bases, quals, read_start, read_stop, mapq, read_skip, ref_off, ref_bases, ref_origin, truth);
	for (auto &x : th) x.join();
	return 0;
}

// The read bases 4 bits each the way a BAM record holds them (what a stager memcpy's out of bam_get_seq): read i from byte
// (read_off[i] >> 1) + i of `out`, first base in the high nibble, codes "=ACMGRSVTWYHKDBN".  Returns -1 if a base has no code.
int ihp_synth_pack4(const uint8_t *bases, const int64_t *read_off, int64_t n_reads, uint8_t *out)
{
	static const char codes[] = "=ACMGRSVTWYHKDBN";
	int8_t lut[256];
	for (int i = 0; i < 256; ++i) lut[i] = -1;
	for (int i = 0; i < 16; ++i) lut[(unsigned char)codes[i]] = (int8_t)i;
	int bad = 0;
	for (int64_t i = 0; i < n_reads; ++i) {
		const uint8_t *s = bases + read_off[i];
		const int64_t len = read_off[i + 1] - read_off[i];
		uint8_t *o = out + (read_off[i] >> 1) + i;
		for (int64_t j = 0; j < len; j += 2) {
			const int a = lut[s[j]], b = j + 1 < len ? lut[s[j + 1]] : 0;
			if (a < 0 || b < 0) bad = 1;
			o[j >> 1] = (uint8_t)(((a & 15) << 4) | (b & 15));
		}
	}
	return bad ? -1 : 0;
}

// The read bases 2 bits each in the library's packed form (ihp_slab2_layout under IHP_SLAB2_BASES_2BIT): read i from 32-bit word
// (read_off[i] >> 4) + i of `out`, sixteen bases per word, base j in bits 2 (j & 15), code (ASCII >> 1) & 3.  Returns -1 (and
// writes nothing useful) if a base is not upper-case A C G T.
int ihp_synth_pack2(const uint8_t *bases, const int64_t *read_off, int64_t n_reads, uint32_t *out)
{
	int bad = 0;
	for (int64_t i = 0; i < n_reads; ++i) {
		const uint8_t *s = bases + read_off[i];
		const int64_t len = read_off[i + 1] - read_off[i];
		uint32_t *o = out + (read_off[i] >> 4) + i;
		for (int64_t j = 0; j < len; j += 16) {
			uint32_t w = 0;
			const int64_t m = len - j < 16 ? len - j : 16;
			for (int64_t k = 0; k < m; ++k) {
				const uint8_t c = s[j + k];
				if (c != 'A' && c != 'C' && c != 'G' && c != 'T') bad = 1;
				w |= (uint32_t)((c >> 1) & 3) << (2 * k);
			}
			o[j >> 4] = w;
		}
	}
	return bad ? -1 : 0;
}

}  // extern "C"
