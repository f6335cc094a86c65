"""py3.10 stand-in for the stdlib tomllib the reference imports (fixture generation only)."""
from tomli import TOMLDecodeError, load, loads  # noqa: F401
