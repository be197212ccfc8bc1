#!/usr/bin/env python3
"""Soak runs under load (round 6): every Winograd kernel family launched hundreds of times back to back while a second stream runs another
convolution, a third streams 256 MB, and the host synchronises every seventh launch (cold restarts) - every result must be bit-identical to the
first.  What found the inline-assembly request bug of the F(4x4,3x3) weight gradient.    python tools/soak_under_load.py [wgrad|fwd]"""
import sys
which = sys.argv[1] if len(sys.argv) > 1 else 'wgrad'
exec(open(__file__.replace('soak_under_load.py', 'soak_%s.py' % which)).read())
