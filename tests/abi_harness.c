/* abi_harness.c -- the C ABI of include/indelope_hip.h seen by a C compiler (gcc), not by ctypes.
 *
 *   gcc -std=c99 -Wall -I include tests/abi_harness.c -L indelope_amd/lib -lindelope_hip -Wl,-rpath,... -o tests/abi_harness.bin
 *
 * Without arguments: prints one JSON object with sizeof / offsetof of every struct that crosses the ABI (the Python
 * side compares them with its ctypes mirrors) and the results of the host-only entry points (no GPU needed).
 * With "--gpu": also drives the device through the ABI the way a C (or Nim) caller would: the reference's own ksw2
 * known-answer pair through the drop-in symbol ksw_extz2_sse (ksw2.nim:171-214), and one small region through
 * ihp_run_regions, checked against the hand-derived expectation of tests/hand_vectors.py (order_decides_contigs), and the
 * end-of-job gather (ihp_dist_*: RCCL behind the C ABI) in a group of one.
 * With "--dist RANK WORLD IDFILE [DEVICE]": one rank of a multi-process gather, no Python anywhere (see dist_rank_main).
 */
#define _DEFAULT_SOURCE          /* usleep (the --dist mode waits for the id file) */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "indelope_hip.h"

#define SZ(T) printf("\"sizeof_" #T "\": %zu, ", sizeof(T))
#define OFF(T, f) printf("\"" #T "." #f "\": %zu, ", offsetof(T, f))

static const char *TGT = "CGAAACTGGGCTACTCCATGACCAGGGGCAAAATAGGCTTTTAGCCGCTGCGTTCTGGGAGCTCCTCCCCCTTCTGGGAGCTCCTCCCCCTCCCCAGAAGGCCAAGGGATGTGGGGGCTGGGGGACTGGGAGGCCTGGCAGTCTT";
static const char *QRY = "CGAAACTGGGCTACTCCATGACCAGGGGCAAAATAGGCTTTTAGCCGCTGCGTTCTGGGAGCTCCTCCCCCTCCCCAGAAGGCCAAGGGATGTTGGGG";

static int gpu_checks(void)
{
	int rc = ihp_init(0);
	if (rc) { fprintf(stderr, "ihp_init: %s / %s\n", ihp_strerror(rc), ihp_last_hip_error()); return 1; }
	/* ksw2.nim:171-214 through the reference's own symbol */
	size_t tl = strlen(TGT), ql = strlen(QRY);
	uint8_t *t = malloc(tl), *q = malloc(ql);
	int8_t mat[25];
	ihp_encode((const uint8_t *)TGT, (int64_t)tl, t);
	ihp_encode((const uint8_t *)QRY, (int64_t)ql, q);
	ihp_matrix(1, -2, mat);
	ksw_extz_t ez;
	memset(&ez, 0, sizeof(ez));
	ksw_extz2_sse(NULL, (int)ql, q, (int)tl, t, 5, mat, 3, 1, -1, -1, KSW_EZ_EXTZ_ONLY | KSW_EZ_RIGHT, &ez);
	if (ihp_ksw_last_status() != 0) { fprintf(stderr, "ksw_extz2_sse failed: %s\n", ihp_last_hip_error()); return 1; }
	/* Ez.cigar (ksw2.nim:22-33): 72M 19D 26M, qstop 98, tstop 117, mqe_t 116 */
	uint32_t off = 0, n = 0, ops[8];
	for (int i = 0; i < ez.n_cigar && n < 8; ++i) {
		if (off >= (uint32_t)ez.max_q) break;
		if ((ez.cigar[i] & 0xf) != 2) off += ez.cigar[i] >> 4;
		ops[n++] = ez.cigar[i];
	}
	if (!(n == 3 && ops[0] == (72u << 4 | 0) && ops[1] == (19u << 4 | 2) && ops[2] == (26u << 4 | 0) && ez.max_q + 1 == 98 &&
	      ez.max_t + 1 == 117 && ez.mqe_t == 116)) {
		fprintf(stderr, "ksw2 KAT mismatch: n=%u max_q=%d max_t=%d mqe_t=%d\n", n, ez.max_q, ez.max_t, ez.mqe_t);
		return 1;
	}
	/* the cigar buffer is the callee's realloc, retained by the caller (ksw2_extz2_sse.c:31-41): a second call reuses it */
	uint32_t *before = ez.cigar; int m_before = ez.m_cigar;
	ksw_extz2_sse(NULL, (int)ql, q, (int)tl, t, 5, mat, 4, 1, 50, 400, 0, &ez);
	if (ez.m_cigar < m_before || (ez.m_cigar == m_before && ez.cigar != before) || ez.n_cigar != 4 || ez.max_q != 71) {
		fprintf(stderr, "ksw2 production call: n_cigar %d max_q %d\n", ez.n_cigar, ez.max_q);
		return 1;
	}
	free(ez.cigar); free(t); free(q);
	/* one region through the batched path: tests/hand_vectors.py order_decides_contigs, order A B C */
	const char *S = "GATTACAGGCTC", *A = "TTGACCTAGATTACAGGCTC", *B = "CAGTGGATGATTACAGGCTC", *C = "GATTACAGGCTCAAGCTTGC";
	uint8_t bases[60], quals[60], mapq[3] = {60, 60, 60}, ref[40];
	memcpy(bases, A, 20); memcpy(bases + 20, B, 20); memcpy(bases + 40, C, 20);
	memset(quals, 30, 60); memset(ref, 'A', 40);
	int64_t rro[2] = {0, 3}, ro[4] = {0, 20, 40, 60}, rs[3] = {100, 100, 108}, re[3] = {120, 120, 128}, fo[2] = {0, 40}, org[1] = {90};
	ihp_params p;
	ihp_params_default(&p);
	p.min_overlap_pct = 0.5;
	ihp_batch_in in;
	memset(&in, 0, sizeof(in));
	in.n_regions = 1; in.n_reads = 3; in.region_read_off = rro; in.read_off = ro; in.bases = bases; in.quals = quals;
	in.read_start = rs; in.read_stop = re; in.mapq = mapq; in.ref_off = fo; in.ref_bases = ref; in.ref_origin = org;
	ihp_batch_out out;
	rc = ihp_run_regions(&p, &in, &out);
	if (rc) { fprintf(stderr, "ihp_run_regions: %s / %s\n", ihp_strerror(rc), ihp_last_hip_error()); return 1; }
	int ok = out.n_regions == 1 && out.n_contigs == 2 && out.n_contigs_pre[0] == 2 && out.status[0] == 0 &&
	         out.ctg_start[0] == 108 && out.ctg_nreads[0] == 2 && out.ctg_seq_off[1] == 12 && memcmp(out.ctg_seq, S, 12) == 0 &&
	         out.ctg_start[1] == 100 && out.ctg_nreads[1] == 1 && out.ctg_seq_off[2] == 32 && memcmp(out.ctg_seq + 12, B, 20) == 0;
	for (int i = 0; ok && i < 12; ++i) ok = out.ctg_support[i] == 2;
	for (int i = 12; ok && i < 32; ++i) ok = out.ctg_support[i] == 1;
	ihp_free_out(&out);
	if (!ok) { fprintf(stderr, "region result differs from the hand-derived expectation\n"); return 1; }
	/* the end-of-job gather (indelope.nim:601-608 needs every region's records in region order on one rank) in a group of ONE:
	 * ncclCommInitRank, the records and the packed results through ihp_dist_*, against what the batch itself hands out */
	{
		uint8_t id[IHP_DIST_ID_BYTES];
		ihp_dist *d = NULL;
		ihp_batch *b = NULL;
		if ((rc = ihp_dist_unique_id(id, sizeof(id))) || (rc = ihp_dist_init(0, 1, id, sizeof(id), &d))) {
			fprintf(stderr, "ihp_dist_init: %s / %s\n", ihp_strerror(rc), ihp_last_hip_error()); return 1;
		}
		if (ihp_dist_rank(d) != 0 || ihp_dist_world(d) != 1) { fprintf(stderr, "ihp_dist_rank / world\n"); return 1; }
		if ((rc = ihp_batch_upload(&p, &in, &b)) || (rc = ihp_batch_run(b))) { fprintf(stderr, "batch: %s\n", ihp_strerror(rc)); return 1; }
		ihp_region_summary got[2], want[1];
		int64_t n_total = -1, counts[1] = {-1};
		memset(got, 0xff, sizeof(got));
		if ((rc = ihp_dist_gather_summaries(d, b, 0, NULL, got, 2, &n_total, counts)) || (rc = ihp_batch_summary_host(b, want, 1))) {
			fprintf(stderr, "ihp_dist_gather_summaries: %s / %s\n", ihp_strerror(rc), ihp_last_hip_error()); return 1;
		}
		if (n_total != 1 || counts[0] != 1 || memcmp(got, want, sizeof(want)) != 0 || want[0].n_contigs != 2 || want[0].n_contigs_pre != 2) {
			fprintf(stderr, "gathered records differ from ihp_batch_summary_host\n"); return 1;
		}
		if (ihp_dist_gather_summaries(d, b, 0, NULL, got, 0, &n_total, counts) != IHP_E_CAPACITY) { fprintf(stderr, "short buffer accepted\n"); return 1; }
		ihp_batch_out outs[1];
		int64_t nbytes[1] = {0};
		if ((rc = ihp_dist_gather_payload(d, b, 0, outs, nbytes))) { fprintf(stderr, "ihp_dist_gather_payload: %s / %s\n", ihp_strerror(rc), ihp_last_hip_error()); return 1; }
		ok = nbytes[0] > 0 && outs[0].n_regions == 1 && outs[0].n_contigs == 2 && outs[0].ctg_start[0] == 108 && memcmp(outs[0].ctg_seq, S, 12) == 0 &&
		     memcmp(outs[0].ctg_seq + 12, B, 20) == 0 && outs[0].ctg_support[0] == 2 && outs[0].ctg_support[12] == 1;
		ihp_free_out(&outs[0]);
		ihp_batch_free(b);
		if ((rc = ihp_dist_finalize(d))) { fprintf(stderr, "ihp_dist_finalize: %s\n", ihp_strerror(rc)); return 1; }
		if (!ok) { fprintf(stderr, "gathered payload differs from the hand-derived expectation\n"); return 1; }
	}
	/* error conventions: bad struct_size -> IHP_E_ARG, nothing allocated */
	p.struct_size = 4;
	if (ihp_run_regions(&p, &in, &out) != IHP_E_ARG) { fprintf(stderr, "bad struct_size accepted\n"); return 1; }
	ihp_shutdown();
	return 0;
}

/* "--dist RANK WORLD IDFILE [DEVICE]": one rank of a multi-process gather with no Python and no torch anywhere -- what a C or Nim
 * launcher does.  Rank 0 makes the id and leaves it in IDFILE (written under another name and renamed), the others wait for
 * the file.  Every rank runs `nreg` copies of the hand-derived region above with its window origin moved by 1000 * rank (so
 * that the root can tell whose records it holds), rank r holding r + 2 regions; the root prints what it gathered as JSON.    */
#include <unistd.h>
static int dist_rank_main(int rank, int world, const char *idfile, int device)
{
	int rc = ihp_init(device);
	if (rc) { fprintf(stderr, "ihp_init(%d): %s / %s\n", device, ihp_strerror(rc), ihp_last_hip_error()); return 2; }
	uint8_t id[IHP_DIST_ID_BYTES];
	if (rank == 0) {
		if ((rc = ihp_dist_unique_id(id, sizeof(id)))) { fprintf(stderr, "ihp_dist_unique_id: %s / %s\n", ihp_strerror(rc), ihp_last_hip_error()); return 2; }
		char tmp[1024];
		snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
		FILE *f = fopen(tmp, "wb");
		if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id) || fclose(f) || rename(tmp, idfile)) { fprintf(stderr, "cannot write %s\n", idfile); return 2; }
	} else {
		FILE *f = NULL;
		for (int i = 0; i < 6000 && !(f = fopen(idfile, "rb")); ++i) usleep(10000);
		if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "no id in %s\n", idfile); return 2; }
		fclose(f);
	}
	ihp_dist *d = NULL;
	if ((rc = ihp_dist_init(rank, world, id, sizeof(id), &d))) {
		/* (RCCL refuses two ranks on one device: the caller of this harness on a one-GPU box reads the text) */
		printf("{\"rank\": %d, \"dist_init\": %d, \"error\": \"%s\"}\n", rank, rc, ihp_last_hip_error());
		return 3;
	}
	const char *A = "TTGACCTAGATTACAGGCTC", *B = "CAGTGGATGATTACAGGCTC", *C = "GATTACAGGCTCAAGCTTGC";
	enum { MAXR = 16 };
	const int nreg = rank + 2;
	if (nreg > MAXR || world > MAXR - 2) return 2;
	uint8_t bases[60 * MAXR], quals[60 * MAXR], mapq[3 * MAXR], ref[40 * MAXR];
	int64_t rro[MAXR + 1], ro[3 * MAXR + 1], rs[3 * MAXR], re[3 * MAXR], fo[MAXR + 1], org[MAXR];
	memset(quals, 30, sizeof(quals)); memset(ref, 'A', sizeof(ref)); memset(mapq, 60, sizeof(mapq));
	for (int r = 0; r < nreg; ++r) {
		/* region r of this rank: reads A B C (two contigs), or -- odd r -- A C only (one contig of two reads and ... see below) */
		memcpy(bases + 60 * r, A, 20); memcpy(bases + 60 * r + 20, B, 20); memcpy(bases + 60 * r + 40, C, 20);
		const int64_t base = 1000 * (int64_t)rank + 10 * r;
		rro[r] = 3 * r; fo[r] = 40 * r; org[r] = 90 + base;
		for (int k = 0; k < 3; ++k) { ro[3 * r + k] = 60 * r + 20 * k; rs[3 * r + k] = 100 + base + (k == 2 ? 8 : 0); re[3 * r + k] = rs[3 * r + k] + 20; }
	}
	rro[nreg] = 3 * nreg; fo[nreg] = 40 * nreg; ro[3 * nreg] = 60 * nreg;
	ihp_params p;
	ihp_params_default(&p);
	p.min_overlap_pct = 0.5;
	ihp_batch_in in;
	memset(&in, 0, sizeof(in));
	in.n_regions = nreg; in.n_reads = 3 * nreg; in.region_read_off = rro; in.read_off = ro; in.bases = bases; in.quals = quals;
	in.read_start = rs; in.read_stop = re; in.mapq = mapq; in.ref_off = fo; in.ref_bases = ref; in.ref_origin = org;
	ihp_batch *b = NULL;
	if ((rc = ihp_batch_upload(&p, &in, &b)) || (rc = ihp_batch_run(b))) { fprintf(stderr, "rank %d batch: %s / %s\n", rank, ihp_strerror(rc), ihp_last_hip_error()); return 2; }
	ihp_region_summary got[MAXR * MAXR];
	int64_t n_total = 0, counts[MAXR];
	if ((rc = ihp_dist_gather_summaries(d, b, 0, NULL, got, MAXR * MAXR, &n_total, counts))) { fprintf(stderr, "rank %d gather: %s / %s\n", rank, ihp_strerror(rc), ihp_last_hip_error()); return 2; }
	ihp_batch_out outs[MAXR];
	int64_t nbytes[MAXR];
	if ((rc = ihp_dist_gather_payload(d, b, 0, outs, nbytes))) { fprintf(stderr, "rank %d payload: %s / %s\n", rank, ihp_strerror(rc), ihp_last_hip_error()); return 2; }
	int bad = 0;
	if (rank == 0) {
		int64_t want_total = 0;
		for (int r = 0; r < world; ++r) { want_total += r + 2; if (counts[r] != r + 2) bad = 1; }
		if (n_total != want_total) bad = 1;
		for (int64_t i = 0; i < n_total; ++i) if (got[i].status != 0 || got[i].n_contigs != 2 || got[i].n_contigs_pre != 2) bad = 1;
		for (int r = 0; r < world; ++r) {
			/* rank order = region order: the contig starts carry the rank */
			if (outs[r].n_regions != r + 2 || outs[r].n_contigs != 2 * (r + 2) || outs[r].ctg_start[0] != 108 + 1000 * (int64_t)r ||
			    outs[r].ctg_start[1] != 100 + 1000 * (int64_t)r) bad = 1;
			ihp_free_out(&outs[r]);
		}
		printf("{\"rank\": 0, \"world\": %d, \"dist_init\": 0, \"n_total\": %lld, \"ok\": %s}\n", world, (long long)n_total, bad ? "false" : "true");
	}
	ihp_batch_free(b);
	rc = ihp_dist_finalize(d);
	ihp_shutdown();
	return bad || rc ? 1 : 0;
}

int main(int argc, char **argv)
{
	if (argc >= 5 && strcmp(argv[1], "--dist") == 0) return dist_rank_main(atoi(argv[2]), atoi(argv[3]), argv[4], argc > 5 ? atoi(argv[5]) : 0);
	/* the device checks first: librccl prints a banner on stdout when it starts, and the JSON object has to stay one (the last) line */
	const int want_gpu = argc > 1 && strcmp(argv[1], "--gpu") == 0;
	const int gpu_rc = want_gpu ? gpu_checks() : 0;
	fflush(stdout);
	printf("{");
	SZ(ksw_extz_t); OFF(ksw_extz_t, max_q); OFF(ksw_extz_t, mqe); OFF(ksw_extz_t, mte); OFF(ksw_extz_t, score);
	OFF(ksw_extz_t, m_cigar); OFF(ksw_extz_t, n_cigar); OFF(ksw_extz_t, cigar);
	SZ(ihp_ez); SZ(ihp_contig); OFF(ihp_contig, len); OFF(ihp_contig, start);
	SZ(ihp_correction); SZ(ihp_match); OFF(ihp_match, corrections); OFF(ihp_match, corr_cap);
	SZ(ihp_genotype_t); OFF(ihp_genotype_t, gl);
	SZ(ihp_params); OFF(ihp_params, min_overlap_pct); OFF(ihp_params, K); OFF(ihp_params, match); OFF(ihp_params, bw);
	OFF(ihp_params, error); OFF(ihp_params, fallback); OFF(ihp_params, fb_match); OFF(ihp_params, fb_flag);
	SZ(ihp_batch_in); OFF(ihp_batch_in, n_reads); OFF(ihp_batch_in, quals); OFF(ihp_batch_in, ref_origin); OFF(ihp_batch_in, trim_hi);
	SZ(ihp_event); OFF(ihp_event, len); OFF(ihp_event, cf_offset); OFF(ihp_event, ref_kmer); OFF(ihp_event, gt); OFF(ihp_event, gl);
	OFF(ihp_event, qual);
	SZ(ihp_batch_out); OFF(ihp_batch_out, n_contigs); OFF(ihp_batch_out, status); OFF(ihp_batch_out, events); OFF(ihp_batch_out, alt_hit);
	SZ(ihp_variant); OFF(ihp_variant, start); OFF(ihp_variant, ad); OFF(ihp_variant, lo); OFF(ihp_variant, ref_off); OFF(ihp_variant, alt_kmer);
	SZ(ihp_variants); SZ(ihp_roi_in); OFF(ihp_roi_in, origin); OFF(ihp_roi_in, max_read_coverage); SZ(ihp_roi_out);
	SZ(ihp_region_summary);
	/* host-only entry points */
	ihp_params p;
	ihp_params_default(&p);
	printf("\"params_default\": [%d, %d, %d, %d, %d, %d, %d, %d], ", p.struct_size, p.K, p.bw, p.zdrop, p.min_reads, p.min_ctg_len,
	       p.combine_min_overlap, p.fb_gap_open);
	int8_t mat[25];
	ihp_matrix(1, -2, mat);
	printf("\"matrix\": [");
	for (int i = 0; i < 25; ++i) printf("%d%s", mat[i], i < 24 ? ", " : "], ");
	uint8_t enc[10];
	ihp_encode((const uint8_t *)"ACGTNacgtn", 10, enc);
	printf("\"encode\": [");
	for (int i = 0; i < 10; ++i) printf("%d%s", enc[i], i < 9 ? ", " : "], ");
	ihp_genotype_t g;
	ihp_genotype(10, 10, 1e-4, &g);               /* genotyper.nim:50-53: HET */
	printf("\"genotype_10_10\": %d, ", g.gt);
	ihp_genotype(0, 0, 1e-4, &g);                 /* :65-67: UNKNOWN */
	printf("\"genotype_0_0\": %d, ", g.gt);
	printf("\"strerror_capacity\": \"%s\", \"version\": \"%s\"", ihp_strerror(IHP_E_CAPACITY), ihp_version());
	if (want_gpu) printf(", \"gpu_checks\": %s", gpu_rc ? "false" : "true");
	printf("}\n");
	return gpu_rc;
}
