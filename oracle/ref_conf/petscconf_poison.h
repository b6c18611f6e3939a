/* PetscDefined() misuse checks of the reference's configure are not reproduced here */
