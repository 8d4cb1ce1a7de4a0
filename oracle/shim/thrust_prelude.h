/* Build shim, force-included (-include): Thrust 1.x pulled these algorithm headers in
 * transitively; rocThrust 2.x does not, and the reference never includes them itself. */
#ifndef ORACLE_SHIM_THRUST_PRELUDE_H_
#define ORACLE_SHIM_THRUST_PRELUDE_H_
#ifdef __cplusplus
#include <thrust/copy.h>
#include <thrust/for_each.h>
#include <thrust/functional.h>
#include <thrust/merge.h>
#include <thrust/reduce.h>
#include <thrust/remove.h>
#include <thrust/scan.h>
#include <thrust/scatter.h>
#include <thrust/sequence.h>
#include <thrust/sort.h>
#include <thrust/transform.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/iterator/discard_iterator.h>
#include <thrust/iterator/transform_iterator.h>
#endif
#endif
