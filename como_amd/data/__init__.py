"""Sequence readers and ground-truth converters: everything either side of the hot path that
only matters once real TUM / Replica / ScanNet data and the DepthCov checkpoint are present.  Host-side file handling; the
frames they return are what `como_amd.odom.sequential.ComoSeq.iter` takes."""
