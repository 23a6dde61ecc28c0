"""Backend switch, in the reference's idiom (/root/reference/src/CSparse3/__config__.py
holds one flag, NATIVE, read once when csc.py is imported: csc.py:34).

BACKEND = "hip" is the only numeric backend of this package: the factor/solve
kernels run on the MI355X through libcsparse3_hip.so and there is no CPU
fallback.  The environment variable CS3_BACKEND, if set, must also say "hip";
any other value makes the import fail loudly instead of silently degrading
(the reference's bare `except:` at csc.py:38 is deliberately not mirrored).
"""
import os

NATIVE = True
BACKEND = os.environ.get("CS3_BACKEND", "hip")
