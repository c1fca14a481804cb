// Core/Geometry/KDTreeSearchParam.h -- how neighbours are chosen
// (shape of O3D/Core/Geometry/KDTreeSearchParam.h:31-80): the k nearest, everything
// inside a radius, or the k nearest inside a radius.
#pragma once

namespace open3d {

class KDTreeSearchParam {
public:
    enum class SearchType { Knn = 0, Radius = 1, Hybrid = 2 };
    virtual ~KDTreeSearchParam() {}
    SearchType GetSearchType() const { return search_type_; }

protected:
    explicit KDTreeSearchParam(SearchType type) : search_type_(type) {}

private:
    SearchType search_type_;
};

class KDTreeSearchParamKNN : public KDTreeSearchParam {
public:
    KDTreeSearchParamKNN(int knn = 30) : KDTreeSearchParam(SearchType::Knn), knn_(knn) {}
    int knn_;
};

class KDTreeSearchParamRadius : public KDTreeSearchParam {
public:
    KDTreeSearchParamRadius(double radius) : KDTreeSearchParam(SearchType::Radius), radius_(radius) {}
    double radius_;
};

class KDTreeSearchParamHybrid : public KDTreeSearchParam {
public:
    KDTreeSearchParamHybrid(double radius, int max_nn)
        : KDTreeSearchParam(SearchType::Hybrid), radius_(radius), max_nn_(max_nn) {}
    double radius_;
    int max_nn_;
};

}  // namespace open3d
