/* placeholder until ipm_ref.c lands */
