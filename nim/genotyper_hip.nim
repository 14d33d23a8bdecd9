## genotyper_hip.nim -- src/genotyper.nim's surface (`GT`, `Genotype`, `genotype`, `qual`, `$`) over ihp_genotype /
## ihp_genotype_qual of include/indelope_hip.h (SOURCE ONLY: no Nim toolchain in the build image).  The arithmetic is
## three fp64 logs per event and stays on the host inside the library, exactly as genotyper.nim:36-47 computes it; in the
## batched path every tallied ihp_event already carries gt / gl / qual.
import strutils
import indelope_hip

type
  GT* {.pure.} = enum                    # genotyper.nim:7-12; the values are IHP_GT_*
    HOM_REF
    HET
    HOM_ALT
    UNKNOWN

  Genotype* = tuple[GT: GT, GL: array[3, float64]]        # genotyper.nim:14

const gl_precision = 4
const gt_encodings = ["0/0", "0/1", "1/1", "./."]

proc `$`*(g: GT): string = gt_encodings[g.int]            # genotyper.nim:19

proc genotype*(r: int, a: int, error: float64): Genotype =   # genotyper.nim:36
  var g: IhpGenotype
  discard ihp_genotype(int64(r), int64(a), error, addr g)
  return (GT: GT(g.gt), GL: g.gl)

proc qual*(g: Genotype): float64 =                        # genotyper.nim:22
  var c: IhpGenotype
  c.gt = int32(g.GT.int); c.gl = g.GL
  return ihp_genotype_qual(addr c)

proc `$`*(g: Genotype): string =                          # genotyper.nim:31
  return ($g.GT &
          ":" & formatFloat(g.qual, precision = gl_precision, format = ffDecimal) &
          ":" & formatFloat(g.GL[0], format = ffDecimal, precision = gl_precision) & "," &
          formatFloat(g.GL[1], format = ffDecimal, precision = gl_precision) & "," &
          formatFloat(g.GL[2], format = ffDecimal, precision = gl_precision))

when isMainModule:                       # genotyper.nim:49-67
  var r = genotype(20 - 10, 10, 1e-4)
  assert r.GT == GT.HET and r.GL[1] > r.GL[0]
  assert genotype(20, 0, 1e-4).GT == GT.HOM_REF
  assert genotype(1, 19, 1e-2).GT == GT.HOM_ALT
  assert genotype(1, 19, 1e-8).GT == GT.HET
  assert genotype(0, 0, 1e-8).GT == GT.UNKNOWN
  assert $genotype(1, 19, 1e-8).GT == "0/1"
