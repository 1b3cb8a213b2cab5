"""Drop-in import path of the reference (``lib.blur``)."""
