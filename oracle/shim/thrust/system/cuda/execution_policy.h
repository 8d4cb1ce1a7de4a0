/* Build shim: intentionally empty. The reference includes this header
 * (query/algorithm.hpp:21, query/transform.hpp:22) but in QUERY_MODE=HOST it only
 * ever uses thrust::host (query/utils.hpp:56-58). */
