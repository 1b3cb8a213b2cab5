"""Drop-in import path of the reference (``lib.*``); implementation in structure_from_motion_amd."""
