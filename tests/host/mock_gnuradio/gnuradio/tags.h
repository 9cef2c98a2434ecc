#ifndef MOCK_GR_TAGS_H
#define MOCK_GR_TAGS_H
#include "gnuradio/block.h"
#endif
