"""CPU oracle for the GoMAvatar hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; nothing under gomavatar_amd/ does.  See
oracle/raster_oracle.c and oracle/geometry.py for what is restated and how
each part is pinned.
"""
