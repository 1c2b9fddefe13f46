"""Mirror of the reference's rec/io package for the index streams the beam-search coder emits."""
from .entropy_coding import ArithmeticCoder  # noqa: F401
from .utils import write_compressed_code, read_compressed_code, encode_files, decode_files  # noqa: F401
