/*
 * oracle_variants.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of what follows the tally inside callsemble (src/indelope.nim:375-428: filters, qual
 * scalings, INFO fields, REF/ALT alleles), get_min_flank (:119-132), mean/median (:146-155), `$`(Variant)
 * (:104-113), `$`(Genotype) (src/genotyper.nim:31-34) and the last-two-variants dedupe of the main loop
 * (:604-608), over one batch's inputs and flat results.
 *
 * PARITY UNPINNED: no reference test covers this part and the Nim program cannot be built here.  AKE/RKE and
 * the :412 filter use the third-party `kmer` package's distance `d` (indelope.nimble:10-11, un-vendored, no
 * version), taken here as the distance of the k-mer window from the closer end of the read.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

typedef struct { char *p; int64_t n, cap; } strbuf;

static int64_t sb_add(strbuf *s, const char *src, int64_t n)
{
	if (s->n + n + 1 > s->cap) {
		s->cap = (s->n + n + 1) * 2 + 256;
		s->p = (char *)realloc(s->p, (size_t)s->cap);
	}
	const int64_t at = s->n;
	memcpy(s->p + at, src, (size_t)n);
	s->n += n;
	return at;
}

/* Ez.cigar (ksw2.nim:22-33): number of leading ops the iterator yields */
static int n_truncated(const uint32_t *cig, int n_cigar, int max_q)
{
	uint32_t max_off = (uint32_t)max_q, off = 0;
	int k = 0;
	for (; k < n_cigar; ++k) {
		if (off >= max_off) break;
		if ((cig[k] & 0xf) != 2) off += cig[k] >> 4;
	}
	return k;
}

/* indelope.nim:119-132 */
static int64_t get_min_flank(int event_type, uint32_t event_len, const uint32_t *cig, int nt)
{
	const int64_t init_len = INT64_MAX;
	int64_t result = init_len;
	int found_event = 0;
	for (int i = 0; i < nt; ++i) {
		const int op = (int)(cig[i] & 0xf);
		const int64_t len = (int64_t)(cig[i] >> 4);
		if (op == 0) {
			if (found_event) result = len < result ? len : result;
			else result = len;
			if (found_event) return result;
		} else if (op - 1 == event_type && len == (int64_t)event_len) {
			if (result == init_len) result = 0;
			found_event = 1;
		}
	}
	return 0;
}

static int cmp_u8(const void *a, const void *b) { return (int)*(const uint8_t *)a - (int)*(const uint8_t *)b; }

static int all_same(const char *s, int n)
{
	for (int i = 1; i < n; ++i) if (s[i] != s[0]) return 0;
	return 1;
}

int orc_call_variants(const ihp_params *p, const ihp_batch_in *in, const ihp_batch_out *out, ihp_variants *vars)
{
	if (!p || !in || !out || !vars) return IHP_E_ARG;
	memset(vars, 0, sizeof(*vars));
	int64_t n = 0;
	for (int64_t e = 0; e < out->n_events; ++e) if (out->events[e].status == IHP_EV_TALLIED) n++;
	ihp_variant *V = (ihp_variant *)calloc((size_t)(n ? n : 1), sizeof(ihp_variant));
	strbuf sb = {0, 0, 0};
	const int K = p->K;
	int64_t k = 0;
	int64_t last[2] = {-1, -1};                          /* last_var, last_var2 (:598-599): indices into V */
	uint8_t *mq = 0; int64_t mq_cap = 0;
	for (int32_t r = 0; r < out->n_regions; ++r) {
		const int64_t r0 = in->region_read_off[r], nreads = in->region_read_off[r + 1] - r0;
		const uint8_t *slice = in->ref_bases + in->ref_off[r];
		const int64_t slen = in->ref_off[r + 1] - in->ref_off[r], origin = in->ref_origin[r];
		if (nreads > mq_cap) { mq_cap = nreads; mq = (uint8_t *)realloc(mq, (size_t)mq_cap); }
		for (int64_t c = out->contig_off[r]; c < out->contig_off[r + 1]; ++c) {
			const uint32_t *cig = out->cigar + out->cigar_off[c];
			const int nt = n_truncated(cig, out->aln_ez[c].n_cigar, out->aln_ez[c].max_q);
			const uint8_t *ctg = out->ctg_seq + out->ctg_seq_off[c];
			const int64_t ctg_len = out->ctg_seq_off[c + 1] - out->ctg_seq_off[c];
			for (int64_t e = out->event_off[c]; e < out->event_off[c + 1]; ++e) {
				const ihp_event *ev = &out->events[e];
				if (ev->status != IHP_EV_TALLIED) continue;
				ihp_variant *v = &V[k++];
				v->region = r; v->contig = (int32_t)(c - out->contig_off[r]); v->event = e;
				v->start = ev->tstart; v->gt = ev->gt; v->gq = ev->qual; v->qual = ev->qual;
				v->gl[0] = ev->gl[0]; v->gl[1] = ev->gl[1]; v->gl[2] = ev->gl[2];
				v->ad[0] = ev->ref_support; v->ad[1] = ev->alt_support;
				v->event_type = ev->type; v->amq = v->rmq = -1; v->ake = v->rke = NAN;
				memcpy(v->ref_kmer, ev->ref_kmer, 32); memcpy(v->alt_kmer, ev->alt_kmer, 32);
				const int ref_support = ev->ref_support, alt_support = ev->alt_support, both_found = ev->both_found;
				const int offset = ev->cf_offset;
				if (alt_support < p->min_reads) { v->filter = IHP_VF_LOW_ALT; continue; }                       /* :375 */
				if ((double)alt_support / (double)nreads < 0.1) { v->filter = IHP_VF_LOW_FRAC; continue; }      /* :377 */
				if (ev->gt == IHP_GT_HOM_REF) { v->filter = IHP_VF_HOM_REF; continue; }                         /* :380 */
				const int mn = ref_support < alt_support ? ref_support : alt_support;
				if (offset == 0 && both_found >= (int)(0.75 * (double)mn)) { v->filter = IHP_VF_BOTH_AT_EDGE; continue; }   /* :384 */
				v->dp = (int32_t)nreads;                                                                       /* :386 */
				if (offset < 5) { v->lo = 1; v->qual /= 2.0; }                                                   /* :387-389 */
				if (both_found > 0) { v->bs = both_found; v->qual /= 1.5; } else v->qual *= 2;                   /* :390-394 */
				{                                                                                                /* :395 CC= */
					char tmp[32];
					v->cc_off = sb.n;
					for (int i = 0; i < nt; ++i) {
						const int m = snprintf(tmp, sizeof(tmp), "%u%c", cig[i] >> 4, "MID"[cig[i] & 0xf]);
						sb_add(&sb, tmp, m);
					}
					v->cc_len = (int32_t)(sb.n - v->cc_off);
				}
				v->al = ev->aligned;                                                                             /* :396-397 */
				const int64_t min_flank = get_min_flank(ev->type, ev->len, cig, nt);                             /* :398 */
				const int64_t tspan = ev->tstop - ev->tstart, qspan = ev->qstop - ev->qstart;
				if (min_flank - 1 < (tspan > qspan ? tspan : qspan)) { v->filter = IHP_VF_SMALL_FLANK; continue; }   /* :400 */
				v->mf = (int32_t)min_flank; v->cf = offset; v->nc = out->n_contigs_pre[r];                       /* :401-403 */
				if (offset == 0) v->qual /= 4.0;                                                                 /* :404-405 */
				/* adists / rdists / amapqs / rmapqs (:302-309) from the first-hit windows */
				{
					const int64_t h0 = out->hit_off[e];
					double asum = 0, rsum = 0; int64_t an = 0, rn = 0;
					for (int side = 0; side < 2; ++side) {
						const int32_t *hit = side ? out->alt_hit : out->ref_hit;
						int64_t m = 0;
						for (int64_t i = 0; i < nreads; ++i) {
							const int32_t w = hit[h0 + i];
							if (w < 0) continue;
							const int64_t len = in->read_off[r0 + i + 1] - in->read_off[r0 + i];
							const int64_t d0 = w, d1 = len - K - w;
							const double d = (double)(d0 < d1 ? d0 : d1);
							if (side) { asum += d; an++; } else { rsum += d; rn++; }
							mq[m++] = in->mapq[r0 + i];
						}
						if (m > 0) {                                                                             /* :408-411, median :152-155 */
							qsort(mq, (size_t)m, 1, cmp_u8);
							const int med = mq[(int64_t)((double)m / 2.0)];
							if (side) v->amq = med; else v->rmq = med;
						}
					}
					v->ake = asum / (double)an; v->rke = rsum / (double)rn;                                      /* :146-150: 0/0 = NaN when empty */
				}
				if (v->ake < 5) { v->filter = IHP_VF_KMER_AT_END; continue; }                                    /* :412 */
				if (ev->type == 1) {                                                                             /* :413-415 deletion */
					const int64_t a = ev->tstart - 1 - origin, b = ev->tstop - 1 - origin;                       /* fai.get: end inclusive */
					if (a < 0 || b >= slen || b < a) { v->filter = IHP_VF_OOB; continue; }
					v->ref_off = sb_add(&sb, (const char *)slice + a, b - a + 1); v->ref_len = (int32_t)(b - a + 1);
					v->alt_off = sb_add(&sb, (const char *)slice + a, 1); v->alt_len = 1;
				} else {                                                                                         /* :420-427 insertion */
					const int64_t a = ev->tstart - 1 - origin;
					if (a < 0 || a >= slen || ev->qstart - 1 < 0 || ev->qstop > ctg_len) { v->filter = IHP_VF_OOB; continue; }
					v->ref_off = sb_add(&sb, (const char *)slice + a, 1); v->ref_len = 1;
					v->alt_len = (int32_t)(ev->qstop - (ev->qstart - 1));
					v->alt_off = sb_add(&sb, (const char *)ctg + ev->qstart - 1, v->alt_len);
					const char *alt = sb.p + v->alt_off;
					if (K >= 11 && all_same(alt + 1, v->alt_len - 1) && all_same(ev->alt_kmer + K - 11, 11) &&
					    all_same(ev->ref_kmer + K - 11, 11)) { v->filter = IHP_VF_HOMOPOLYMER; continue; }
				}
				/* main loop :604-608: skip what equals one of the last two printed */
				int dup = 0;
				for (int j = 0; j < 2 && !dup; ++j) {
					if (last[j] < 0) continue;
					const ihp_variant *o = &V[last[j]];
					dup = o->start == v->start && o->ref_len == v->ref_len && o->alt_len == v->alt_len &&
					      memcmp(sb.p + o->ref_off, sb.p + v->ref_off, (size_t)v->ref_len) == 0 &&
					      memcmp(sb.p + o->alt_off, sb.p + v->alt_off, (size_t)v->alt_len) == 0;
				}
				if (dup) { v->filter = IHP_VF_DUPLICATE; continue; }
				v->filter = IHP_VF_EMITTED;
				last[1] = last[0]; last[0] = k - 1;
			}
		}
	}
	free(mq);
	vars->n = k; vars->v = V; vars->n_chars = sb.n; vars->chars = sb.p ? sb.p : (char *)calloc(1, 1);
	return 0;
}

void orc_free_variants(ihp_variants *vars)
{
	if (!vars) return;
	free(vars->v); free(vars->chars);
	memset(vars, 0, sizeof(*vars));
}

/* Nim's formatFloat(x, ffDecimal, precision) of the values that occur here */
static int fmt_float(char *dst, size_t cap, double x, int prec)
{
	if (isnan(x)) return snprintf(dst, cap, "nan");
	if (isinf(x)) return snprintf(dst, cap, x < 0 ? "-inf" : "inf");
	return snprintf(dst, cap, "%.*f", prec, x);
}

int64_t orc_format_variant(const ihp_variant *v, const char *chars, const char *chrom, char *buf, int64_t cap)
{
	static const char *GT[4] = {"0/0", "0/1", "1/1", "./."};
	if (!v || !chars || !chrom) return IHP_E_ARG;
	strbuf s = {0, 0, 0};
	char t[96];
#define ADD(str) sb_add(&s, (str), (int64_t)strlen(str))
#define ADDF(...) do { const int m_ = snprintf(t, sizeof(t), __VA_ARGS__); sb_add(&s, t, m_); } while (0)
	ADD(chrom); ADDF("\t%lld\t.\t", (long long)v->start);                                       /* :108-109 */
	sb_add(&s, chars + v->ref_off, v->ref_len); ADD("\t"); sb_add(&s, chars + v->alt_off, v->alt_len); ADD("\t");
	fmt_float(t, sizeof(t), v->qual, 2); ADD(t); ADD("\tPASS\t");                               /* :104-107, :111 */
	ADDF("AD=%d,%d;ref_kmer=", v->ad[0], v->ad[1]); ADD(v->ref_kmer); ADD(";alt_kmer="); ADD(v->alt_kmer);   /* :62-66 */
	ADDF(";DP=%d", v->dp);
	if (v->lo) ADD(";LO");
	if (v->bs > 0) ADDF(";BS=%d", v->bs);
	ADD(";CC="); sb_add(&s, chars + v->cc_off, v->cc_len);
	if (v->al) ADD(";AL");
	ADDF(";MF=%d;CF=%d;NC=%d;AKE=", v->mf, v->cf, v->nc);
	fmt_float(t, sizeof(t), v->ake, 2); ADD(t); ADD(";RKE="); fmt_float(t, sizeof(t), v->rke, 2); ADD(t);
	if (v->amq >= 0) ADDF(";AMQ=%d", v->amq);
	if (v->rmq >= 0) ADDF(";RMQ=%d", v->rmq);
	ADD("\tGT:GQ:GL\t"); ADD(GT[v->gt & 3]); ADD(":");                                          /* genotyper.nim:31-34 */
	fmt_float(t, sizeof(t), v->gq, 4); ADD(t);
	for (int g = 0; g < 3; ++g) { ADD(g ? "," : ":"); fmt_float(t, sizeof(t), v->gl[g], 4); ADD(t); }
#undef ADD
#undef ADDF
	const int64_t need = s.n;
	if (buf && cap > 0) {
		const int64_t m = need < cap - 1 ? need : cap - 1;
		memcpy(buf, s.p, (size_t)m); buf[m] = 0;
	}
	free(s.p);
	return need;
}
