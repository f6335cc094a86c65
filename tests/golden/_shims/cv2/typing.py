MatLike = object
