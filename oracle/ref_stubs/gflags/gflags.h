// Stand-in for <gflags/gflags.h>: the sources compiled by oracle/Makefile.ref define no flags.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_STUBS_GFLAGS_H_
#define ORACLE_REF_STUBS_GFLAGS_H_
#endif
