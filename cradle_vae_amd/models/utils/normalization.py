from ...guide import LogStandardizationModule  # noqa: F401
