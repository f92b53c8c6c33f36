"""Import shims: the reference's caller does `from util.point_cloud_to import ...` (dpc/models/model_pc_to.py:15-17)
because dpc/run/startup.py puts `dpc/` on sys.path.  Putting THIS package's parent directory on sys.path
instead makes the same imports resolve to the MI355X implementation in dpc.render."""
