#if !defined(INCLUDED_PETSCFIX_H)
#define INCLUDED_PETSCFIX_H
#endif
