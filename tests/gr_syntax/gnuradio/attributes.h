/* syntax-check scaffolding only (tests/gr_syntax/README.md) */
#ifndef GRSYN_ATTRIBUTES_H
#define GRSYN_ATTRIBUTES_H
#define __GR_ATTR_EXPORT
#define __GR_ATTR_IMPORT
#endif
