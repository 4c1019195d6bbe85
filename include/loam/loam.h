/** @brief Convenience header including the whole loam API (MI355X back end). */
#pragma once
#include "common.h"
#include "features.h"
#include "geometry.h"
#include "kdtree.h"
#include "registration.h"
