"""Exception types of the package.

`CodingError` is the reference's error convention for the coding path (rec/coding/utils.py:6): its drivers catch it per
image and move on (examples/lossless/compression_performance.py:346,375-377).  Every failure the C-ABI library reports
while coding (bad shapes, K overflow, HIP errors, missing library / device) is raised as `IrecLibraryError`, which IS a
`CodingError`, so the same `except CodingError` keeps working; it is also a `RuntimeError` for callers that treat a
missing GPU as an environment problem.
"""


class CodingError(Exception):
    """Base exception for errors occurring in irec.coding (reference: rec/coding/utils.py:6)."""


class IrecLibraryError(CodingError, RuntimeError):
    """libirec_hip.so is missing, has no device, or one of its entry points returned a negative irec_status."""
