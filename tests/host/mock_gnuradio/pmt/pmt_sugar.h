#ifndef MOCK_PMT_SUGAR_H
#define MOCK_PMT_SUGAR_H
#include "pmt/pmt.h"
#endif
