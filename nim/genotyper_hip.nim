## genotyper_hip.nim -- a PATCH for src/genotyper.nim, not a module: `include` it after genotyper.nim's type section
## (genotyper.nim:7-14); it adds one proc that computes `genotype` (genotyper.nim:36) through ihp_genotype of
## include/indelope_hip.h.  `GT`, `Genotype`, `qual`, both `$` and the module's own asserts stay the reference's code
## (INTEGRATION.md section 3 shows the diff).  The arithmetic is three fp64 logs per event on the host inside the library;
## in the batched path every tallied ihp_event already carries gt / gl / qual.
## SOURCE ONLY: no Nim toolchain in the build image.
import indelope_hip

proc hip_genotype(r: int, a: int, error: float64): Genotype =
  var g: IhpGenotype
  discard ihp_genotype(int64(r), int64(a), error, addr g)
  result.GT = GT(g.gt)                     # the values of genotyper.nim:7-12 are IHP_GT_*
  result.GL = g.gl
