"""Drop-in import surface: `from vapor_vali.Simple_function import *` resolves to the HIP-backed
implementations in vapor_amd (the reference package of the same name is pure Python/Cython)."""
