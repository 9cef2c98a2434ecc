#ifndef MOCK_GR_TYPES_H
#define MOCK_GR_TYPES_H
#include <vector>
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;
#endif
