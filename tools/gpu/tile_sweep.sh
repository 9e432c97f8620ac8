# Block-tile sweep of the split-operand tile kernels (convgs, conv3d_mfma16.hip) -- a record of the round-4 sweep, not a
# runnable harness: launch_layer_split() was built with one `case` per candidate tile (BZ x BY x BX M-tiles of 2 x 8
# outputs) chosen by an environment variable, and every candidate was timed with
#   python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic
# (stages.conv2/3/4.ms; bench.py does not check results: tools/gpu/split_layers.sh ran on the chosen tiles afterwards).
# cfg2, ms, the tile before the sweep first:
#   conv2 16->16     2x4x2 0.0517 | 4x2x2 0.0516  4x4x1 0.0519  8x1x2 0.0517  8x2x1 0.0526  2x2x2 0.056  4x1x4 0.056  2x2x4 0.056
#                                   2x4x1 0.057  4x2x1 0.057  3x4x2 0.071  4x4x2 0.077  8x4x1 0.079  2x8x1 0.080  2x4x4 0.094
#   conv3 16->32 s2  1x2x2 0.0275 | 2x2x1 0.0238  4x1x1 0.0241  1x4x1 0.0256  3x2x1 0.026  6x1x1 0.026  2x1x2 0.027  1x2x1 0.028
#                                   2x1x1 0.028  2x3x1 0.029  4x2x1 0.030  8x1x1 0.030  1x1x2 0.030  1x1x4 0.033  2x2x2 0.036  1x4x2 0.042
#   conv4 32->32     1x2x2 0.0325 | 4x2x1 0.0250  1x2x4 0.0255  2x4x1 0.0256  8x1x1 0.0257  4x1x1 0.0265  2x2x1 0.0267  6x1x1 0.027
#                                   3x2x1 0.027  1x1x8 0.027  8x2x1 0.032  6x2x1 0.032  4x4x1 0.032  4x3x1 0.033  12x1x1 0.033
#                                   2x2x2 0.036  4x1x2 0.037  4x2x2 0.038  2x4x2 0.038
# chosen: conv2 2x4x2 (unchanged), conv3 2x2x1, conv4 4x2x1
