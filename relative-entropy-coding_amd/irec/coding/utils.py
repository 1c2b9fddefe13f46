class CodingError(Exception):
    """Base exception for errors occurring in irec.coding (reference: rec/coding/utils.py:6)."""
