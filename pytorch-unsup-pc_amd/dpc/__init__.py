"""Namespace mirroring the reference's `dpc/` tree; only the projection path (`dpc.render`) exists here."""
