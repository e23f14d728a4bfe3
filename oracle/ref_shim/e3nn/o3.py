from oracle.adapter_ref import matrix_to_angles, wigner_D  # noqa: F401
