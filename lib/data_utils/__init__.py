"""Drop-in import path of the reference (``lib.data_utils``)."""
